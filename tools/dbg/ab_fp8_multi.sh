# same-box timing of the fused (multi-segment) fp8 kernel on dense-only segments, library variants alternating
# usage: [S=75600 H=8] bash tools/dbg/ab_fp8_multi.sh "" _x ...
for rnd in 1 2; do
for v in "$@"; do
  VORTA_HIP_LIB=vorta_amd/csrc/libvorta_hip$v.so python - <<PY
import os, sys, torch
sys.path.insert(0, os.getcwd())
from vorta_amd import ops
dev = torch.device("cuda:0")
S, H = int(os.environ.get("S", 75600)), int(os.environ.get("H", 8))
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
hl = [torch.arange(0, H // 2, dtype=torch.int32, device=dev), torch.arange(H // 2, H, dtype=torch.int32, device=dev)]
calls = [dict(q=f8.q, k=f8.k, v=f8.v, out=o, n_q=S, n_kv=S, v_descale=f8.v_descale, head_list=h, n_heads=H // 2) for h in hl]
tcalls = [dict(c, q_rows=ident, kv_rows=ident) for c in calls]
a = t(lambda: ops.attn_fwd(f8.q, f8.k, f8.v, o, n_q=S, n_kv=S, v_descale=f8.v_descale))
b = t(lambda: ops.attn_fwd_batch([dict(c) for c in calls]))
c = t(lambda: ops.attn_fwd_batch([dict(c) for c in tcalls]))
print(f"lib{'$v':6s} S={S} H={H}: single {a:.3f} ms | fused, two dense segments {b:.3f} ms | fused, two table segments {c:.3f} ms", flush=True)
PY
done; done
