# alternating same-box timing of fp8 library variants: dense launch + a table launch (identity tables)
# usage: bash tools/dbg/ab_fp8.sh "" _A _C ...   (suffixes of vorta_amd/csrc/libvorta_hip<suffix>.so)
for rnd in 1 2; do
for v in "$@"; do
  VORTA_HIP_LIB=vorta_amd/csrc/libvorta_hip$v.so python - <<PY
import os, sys, torch
sys.path.insert(0, os.getcwd())
from vorta_amd import ops
dev = torch.device("cuda:0")
S, H = 75600, 8
q, k, v = (torch.randn((H, S, 128), device=dev, dtype=torch.bfloat16) for _ in range(3))
o = torch.empty_like(q)
f8 = ops.fp8_quantize_qkv(q, k, v)
ident = torch.arange(S, dtype=torch.int32, device=dev)
def t(fn, n=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
a = t(lambda: ops.attn_fwd(f8.q, f8.k, f8.v, o, n_q=S, n_kv=S, v_descale=f8.v_descale))
b = t(lambda: ops.attn_fwd(f8.q, f8.k, f8.v, o, n_q=S, n_kv=S, v_descale=f8.v_descale, q_rows=ident, kv_rows=ident))
print(f"lib{'$v':6s} dense {a:.3f} ms   tables {b:.3f} ms", flush=True)
PY
done; done
