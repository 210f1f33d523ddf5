# alternating same-box timing of 16-bit library variants: dense launch, table launch, fused two-segment launch
for rnd in 1 2; do
for v in "$@"; do
  VORTA_HIP_LIB=vorta_amd/csrc/libvorta_hip$v.so python - <<PY
import os, sys, torch
sys.path.insert(0, os.getcwd())
from vorta_amd import ops
dev = torch.device("cuda:0")
S, H = 75600, 8
dt = torch.bfloat16 if os.environ.get("DT", "bf16") == "bf16" else torch.float16
q, k, v = (torch.randn((H, S, 128), device=dev, dtype=dt) for _ in range(3))
o = torch.empty_like(q)
ident = torch.arange(S, dtype=torch.int32, device=dev)
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
hl = [torch.arange(0, 4, dtype=torch.int32, device=dev), torch.arange(4, 8, dtype=torch.int32, device=dev)]
calls = [dict(q=q, k=k, v=v, out=o, n_q=S, n_kv=S, head_list=h, n_heads=4) for h in hl]
a = t(lambda: ops.attn_fwd(q, k, v, o, n_q=S, n_kv=S))
b = t(lambda: ops.attn_fwd(q, k, v, o, n_q=S, n_kv=S, q_rows=ident, kv_rows=ident))
c = t(lambda: ops.attn_fwd_batch([dict(x) for x in calls]))
print(f"lib{'$v':6s} dense {a:.3f} ms   tables {b:.3f} ms   fused (two dense segments) {c:.3f} ms", flush=True)
PY
done; done
