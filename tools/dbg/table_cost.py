#!/usr/bin/env python
"""What do the row tables cost the dense launch?  S = 118 800 + 256, H = 3 (one rank of 8), bf16 / fp16 / fp8: no tables,
identity tables, and the Ulysses receive layout's row map."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from vorta_amd import ops


def timeit(fn, n=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


dev = torch.device("cuda:0")
H, S = 3, 118800
ident = torch.arange(S, dtype=torch.int32, device=dev)
flops = 4.0 * S * S * 128 * H
for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
    q, k, v = (torch.randn((H, S, 128), device=dev, dtype=dt) for _ in range(3))
    o = torch.empty_like(q)
    for rnd in range(2):
        a = timeit(lambda: ops.attn_fwd(q, k, v, o, n_q=S, n_kv=S))
        b = timeit(lambda: ops.attn_fwd(q, k, v, o, n_q=S, n_kv=S, q_rows=ident, kv_rows=ident))
        print(f"{name}: no tables {a:.3f} ms {flops / a / 1e9:.0f} TFLOP/s | identity tables {b:.3f} ms {flops / b / 1e9:.0f} "
              f"TFLOP/s ({100 * (b / a - 1):+.1f} %)", flush=True)
    if name == "bf16":
        f8 = ops.fp8_quantize_qkv(q, k, v)
        for rnd in range(2):
            a = timeit(lambda: ops.attn_fwd(f8.q, f8.k, f8.v, o, n_q=S, n_kv=S, v_descale=f8.v_descale))
            b = timeit(lambda: ops.attn_fwd(f8.q, f8.k, f8.v, o, n_q=S, n_kv=S, v_descale=f8.v_descale, q_rows=ident,
                                            kv_rows=ident))
            print(f"fp8 : no tables {a:.3f} ms {flops / a / 1e9:.0f} TFLOP/s | identity tables {b:.3f} ms "
                  f"{flops / b / 1e9:.0f} TFLOP/s ({100 * (b / a - 1):+.1f} %)", flush=True)
