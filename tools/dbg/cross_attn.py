#!/usr/bin/env python
"""Wan text cross-attention (wan.py:142-145 with Sq != Skv): S = 75 600 queries x 512 keys, H = 40 (Wan-14B 81f 720p)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from vorta_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for H, S, Skv in ((40, 75600, 512), (12, 32760, 512), (40, 75600, 257)):
    q = torch.randn((H, S, 128), device=dev, dtype=torch.bfloat16)
    k, v = (torch.randn((H, Skv, 128), device=dev, dtype=torch.bfloat16) for _ in range(2))
    o = torch.empty_like(q)
    flops = 4.0 * S * Skv * 128 * H
    for br in (256, 128):
        ms = timeit(lambda: ops.attn_fwd(q, k, v, o, n_q=S, n_kv=Skv, block_rows=br))
        print(f"H={H} Sq={S} Skv={Skv} block_rows={br}: {ms:.3f} ms  {flops / ms / 1e9:.0f} TFLOP/s; "
              f"q+o traffic {2 * q.numel() * 2 / ms / 1e9:.2f} TB/s", flush=True)
    import torch.nn.functional as F
    ms = timeit(lambda: F.scaled_dot_product_attention(q[None], k[None], v[None]))
    print(f"   torch SDPA: {ms:.3f} ms  {flops / ms / 1e9:.0f} TFLOP/s", flush=True)
