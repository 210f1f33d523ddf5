#!/usr/bin/env python
"""Does the convert pass of the fp8 quantiser run faster when it re-reads a head's q,k,v right after the abs-max pass read
them (91 MB per head: Infinity Cache resident) than after the abs-max pass over ALL heads (5.5 GB)?  Run under
rocprofv3 --kernel-trace --stats and compare the kernels' summed durations of the two phases (marked by the row counts)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from vorta_amd import ops

dev = torch.device("cuda:0")
H, S = 24, 118800 + 256
q, k, v = (torch.randn((H, S, 128), device=dev, dtype=torch.bfloat16) for _ in range(3))
f_all = ops.fp8_quantize_qkv(q, k, v, center_k=True)
f_one = [ops.fp8_quantize_qkv(q[h:h + 1], k[h:h + 1], v[h:h + 1], center_k=True) for h in range(H)]
torch.cuda.synchronize()


def timed(fn, n=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def per_head():
    for h in range(H):
        ops.fp8_quantize_qkv(q[h:h + 1], k[h:h + 1], v[h:h + 1], out=f_one[h], center_k=True)


print(f"all heads in one call: {timed(lambda: ops.fp8_quantize_qkv(q, k, v, out=f_all, center_k=True)):.3f} ms")
print(f"one call per head    : {timed(per_head):.3f} ms (wall, incl. {4 * H} launches)")
