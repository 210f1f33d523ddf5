# same-box timing of the e4m3 kernel, library variants alternating: single dense launch, fused grid of two dense segments,
# fused grid of two table segments (S = 75 600, H = 8)
# usage: VARIANTS="base _x" bash tools/dbg/ab_fp8_lib.sh
for rnd in 1 2; do
for v in ${VARIANTS:-base}; do
  s=$v; [ "$v" = base ] && s=""
  VORTA_HIP_LIB=vorta_amd/csrc/libvorta_hip$s.so TAG=$v python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from vorta_amd import ops
dev = torch.device("cuda:0")
S, H = int(os.environ.get("S", 75600)), int(os.environ.get("H", 8))
q, k, v = (torch.randn((H, S, 128), device=dev, dtype=torch.bfloat16) for _ in range(3))
o = torch.empty_like(q)
f8 = ops.fp8_quantize_qkv(q, k, v, center_k=True)
ident = torch.arange(S, dtype=torch.int32, device=dev)
def t(fn, n=6):
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
fl = 4.0 * S * S * 128 * H / 1e9
print(f"lib {os.environ['TAG']:5s}: single {a:.3f} ms ({fl/a:.0f} TF) | fused dense {b:.3f} ms ({fl/b:.0f} TF) | fused table {c:.3f} ms ({fl/c:.0f} TF)", flush=True)
PY
done; done
