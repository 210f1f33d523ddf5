"""How much accuracy does writing exp2 straight into e4m3 bits cost?  P' = 2^x is replaced by the e4m3 value whose
byte is rint(8 x + 56) (exponent = integer part, mantissa = linear interpolation of the fraction).  Compares, on
the same e4m3 operands, output PSNR (over max|x|) of: exact P, RNE-rounded P, byte-direct P."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import vorta_oracle as vo
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import _fp8_inputs as fi


def attend(Q, K, V, mode, p_bias=5.0):
    z = Q @ K.T
    m = z.max(1, keepdims=True)
    x = z - m + p_bias
    if mode == "exact":
        P = np.exp2(x)
    elif mode == "rne":
        P = vo.e4m3_round(np.exp2(x))
    else:
        b = np.clip(np.rint(8 * x + 56), 0, 126)
        P = vo.e4m3_decode(b.astype(np.int64))
    return (P @ V) / P.sum(1, keepdims=True)


def psnr(a, ref):
    return 10 * np.log10(np.abs(ref).max() ** 2 / np.mean((a - ref) ** 2))


import torch
gen = torch.Generator().manual_seed(0)
D = 128
for fam, q, k, v in fi.families((8, 16, 32), 1, 0, gen, "cpu"):
    to = lambda a: a.to(torch.bfloat16).float().numpy()
    q, k, v = to(q), to(k), to(v)
    kc = k.mean(1)
    qd = vo.fp8_quantize_qkv(q, k, v, k_center=kc)
    Q = vo.e4m3_decode(qd["q8"][0]); K = vo.e4m3_decode(qd["k8"][0]); V = vo.e4m3_decode(qd["v8"][0]) * qd["v_descale"][0]
    ref16 = vo._softmax_attend(q[0].astype(np.float64), k[0].astype(np.float64), v[0].astype(np.float64))
    outs = {m: attend(Q[:512], K, V, m) for m in ("exact", "rne", "direct")}
    # 16-bit q,k (fp8pv): logits exact
    z = (q[0, :512].astype(np.float64) @ k[0].astype(np.float64).T) / np.sqrt(D) * 1.4426950408889634
    def pv(mode):
        m = z.max(1, keepdims=True); x = z - m + 5.0
        if mode == "rne": P = vo.e4m3_round(np.exp2(x))
        else: P = vo.e4m3_decode(np.clip(np.rint(8 * x + 56), 0, 126).astype(np.int64))
        return (P @ V) / P.sum(1, keepdims=True)
    print(f"{fam:18s} fp8: exact {psnr(outs['exact'], ref16[:512]):.2f} rne {psnr(outs['rne'], ref16[:512]):.2f} direct {psnr(outs['direct'], ref16[:512]):.2f}"
          f" | fp8pv: rne {psnr(pv('rne'), ref16[:512]):.2f} direct {psnr(pv('direct'), ref16[:512]):.2f}")
