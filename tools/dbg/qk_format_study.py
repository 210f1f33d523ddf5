#!/usr/bin/env python
"""What would hold 40 dB on peaked logits?  A torch emulation (no kernels of this library) of dense attention with the
score operands in different 8-bit formats, everything else fixed: P and V in e4m3 with the kernel's scaling, fp32
accumulation.  Formats of q, k:
    e4m3/head   per-head multipliers, keys centred (what attn_fwd_fp8.hip computes)
    int8/blk    per-64-token-block abs-max scales, keys centred (7 bits near the block maximum, the same MFMA rate)
    16-bit      q, k as they are (the e4m3 cost of P and V alone)
Inputs: the families of tests/_fp8_inputs.py on a (8, 32, 64) latent (S = 16 384, one head each).  Prints PSNR over max|x| of
the fp32 reference output."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from _fp8_inputs import NAMES, families

dev = torch.device("cuda:0")
latent = (8, 32, 64)
S = latent[0] * latent[1] * latent[2]
gen = torch.Generator(device=dev).manual_seed(1234)
c0 = (1.0 / math.sqrt(128)) * 1.4426950408889634


def e4m3(x):
    return x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()


def attend(q, k, v, p_round):
    """q, k: float32 (S, D) operands whose product is the exp2-domain score; v float32; P rounded to e4m3 (x 32, per 64-key
    block reference = running row max as in the kernel is approximated by the global row max) when p_round"""
    out = torch.empty_like(v)
    for r0 in range(0, q.shape[0], 2048):
        s = q[r0:r0 + 2048] @ k.T  # exp2 domain
        m = s.amax(-1, keepdim=True)
        p = torch.exp2(s - m)
        if p_round:
            p = e4m3(p * 32.0) / 32.0
        out[r0:r0 + 2048] = (p @ v) / p.sum(-1, keepdim=True)
    return out


def quant_e4m3_head(q, k):
    c = k.mean(0, keepdim=True)
    kc = k - c
    aq, ak = q.abs().max(), kc.abs().max()
    t = torch.sqrt(ak / (c0 * aq))
    return e4m3(q * c0 * t), e4m3(kc / t)


def quant_int8_block(q, k, blk=64):
    c = k.mean(0, keepdim=True)
    kc = k - c

    def qi(x):
        xb = x.view(-1, blk, x.shape[-1])
        sc = xb.abs().amax((1, 2), keepdim=True) / 127.0
        return (torch.round(xb / sc).clamp(-127, 127) * sc).view_as(x)
    return qi(q) * c0, qi(kc)


def quant_v(v):
    am = v.abs().amax(0, keepdim=True)
    return e4m3(v * (240.0 / am)) * (am / 240.0)


def psnr(x, ref):
    mse = torch.mean((x - ref) ** 2).item()
    return 10.0 * math.log10(ref.abs().max().item() ** 2 / max(mse, 1e-30))


print(f"dense attention, S = {S}, D = 128, bf16 inputs; PSNR over max|x| (dB) of the fp32 result; P, V in e4m3 in every column")
print(f"{'family':62s} {'e4m3/head':>10s} {'int8/blk64':>11s} {'16-bit q,k':>11s}")
for key, q, k, v in families(latent, 1, 0, gen, dev):
    q, k, v = (x[0].to(torch.bfloat16).float() for x in (q, k, v))
    ref = attend(q * c0, k, v, False)
    v8 = quant_v(v)
    row = []
    for qq, kk in (quant_e4m3_head(q, k), quant_int8_block(q, k), (q * c0, k - k.mean(0, keepdim=True))):
        row.append(psnr(attend(qq, kk, v8, True), ref))
    print(f"{NAMES[key]:62s} {row[0]:10.1f} {row[1]:11.1f} {row[2]:11.1f}", flush=True)
