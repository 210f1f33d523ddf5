#!/usr/bin/env python
"""The same dense launch through the single-launch kernel and through the fused (multi-segment) kernel with ONE segment."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from vorta_amd import _C, ops

dev = torch.device("cuda:0")


def t(fn, n=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


lib = _C.lib()
for S, H in ((75600, 8), (118800, 4), (32760, 12)):
    for dt in (torch.bfloat16, torch.float16):
        q, k, v = (torch.randn((H, S, 128), device=dev, dtype=dt) for _ in range(3))
        o = torch.empty_like(q)
        a, keep = ops._attn_args(q, k, v, o, n_q=S, n_kv=S)
        arr = (_C.AttnArgs * 1)(a)
        for rnd in range(2):
            x = t(lambda: _C.check(lib.vorta_attn_fwd(C.byref(a), ops._stream()), "single"))
            y = t(lambda: _C.check(lib.vorta_attn_fwd_batch(arr, 1, ops._stream()), "batch"))
            print(f"S={S} H={H} {dt}: single-launch kernel {x:.3f} ms | fused kernel, one segment {y:.3f} ms ({100 * (y / x - 1):+.1f} %)",
                  flush=True)
