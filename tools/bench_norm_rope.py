#!/usr/bin/env python
"""Bandwidth of the fused qk-norm + RoPE producer pass (vorta_qk_norm_rope) and of the fp8 quantiser
(vorta_fp8_quantize_qkv) on the Hunyuan-129f tensor shapes: both are HBM-bound, one line each."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vorta_amd import ops


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda:0")
    H, S, T = 24, 118800, 256
    for dt, name in ((torch.bfloat16, "bf16"), (torch.float16, "fp16")):
        x = torch.randn((H, S + T, 128), device=dev, dtype=dt)
        w = torch.randn(128, device=dev, dtype=dt)
        ang = torch.randn((S, 128), device=dev)
        cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
        ms = timeit(lambda: ops.qk_norm_rope(x, w, 1e-6, cos=cos, sin=sin, rope_tokens=S))
        gb = 2 * x.numel() * 2 / 1e9  # one read + one write of the tensor
        print(f"vorta_qk_norm_rope {name} (H={H}, {S}+{T} tokens, per-head norm + rope): {ms:.3f} ms  {gb / ms:.2f} TB/s "
              f"({gb:.2f} GB read+write)", flush=True)
    q, k, v = (torch.randn((H, S + T, 128), device=dev, dtype=torch.bfloat16) for _ in range(3))
    f8 = ops.fp8_quantize_qkv(q, k, v, center_k=True)
    ms = timeit(lambda: ops.fp8_quantize_qkv(q, k, v, out=f8, center_k=True))
    gb = q.numel() * (2 + 3 * 3) / 1e9
    print(f"vorta_fp8_quantize_qkv bf16 (H={H}, {S + T} tokens; abs-max pass over v 2 B, convert pass over q,k,v 2 B in / "
          f"1 B out per element; q/k abs-max from the centre's token sample): {ms:.3f} ms  {gb / ms:.2f} TB/s ({gb:.2f} GB)",
          flush=True)


if __name__ == "__main__":
    main()
