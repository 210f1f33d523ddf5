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
        gb = 2 * x.numel() * 2 / 1e9  # one read + one write of the tensor
        # head-major (H, S, D) and the projection's own token-major layout (S, H*D) viewed as (H, S, D)
        xt = torch.randn((S + T, H, 128), device=dev, dtype=dt).transpose(0, 1)
        for lay, t in (("head-major (H,S,D)", x), ("token-major (S,H*D) view", xt)):
            ms = timeit(lambda: ops.qk_norm_rope(t, w, 1e-6, cos=cos, sin=sin, rope_tokens=S))
            print(f"vorta_qk_norm_rope {name} (H={H}, {S}+{T} tokens, per-head norm + rope), {lay}: {ms:.3f} ms  "
                  f"{gb / ms:.2f} TB/s ({gb:.2f} GB read+write)", flush=True)
    # Wan-2.1 14B: RMSNorm across all 40 heads of a token (weight (H*D,)), 75 600 tokens, the projection's (S, H*D) layout
    Hw, Sw = 40, 75600
    xw = torch.randn((Sw, Hw, 128), device=dev, dtype=torch.bfloat16).transpose(0, 1)
    ww = torch.randn(Hw * 128, device=dev, dtype=torch.bfloat16)
    angw = torch.randn((Sw, 128), device=dev)
    cw, sw = angw.cos().contiguous(), angw.sin().contiguous()
    ms = timeit(lambda: ops.qk_norm_rope(xw, ww, 1e-6, cos=cw, sin=sw, across_heads=True))
    gbw = 2 * xw.numel() * 2 / 1e9
    print(f"vorta_qk_norm_rope bf16 (H={Hw}, {Sw} tokens, norm across heads + rope), token-major (S,H*D) view: {ms:.3f} ms  "
          f"{gbw / ms:.2f} TB/s ({gbw:.2f} GB read+write)", flush=True)
    q, k, v = (torch.randn((H, S + T, 128), device=dev, dtype=torch.bfloat16) for _ in range(3))
    f8 = ops.fp8_quantize_qkv(q, k, v, center_k=True)
    ms = timeit(lambda: ops.fp8_quantize_qkv(q, k, v, out=f8, center_k=True))
    gb = q.numel() * (2 + 3 * 3) / 1e9
    print(f"vorta_fp8_quantize_qkv bf16 (H={H}, {S + T} tokens; abs-max pass over v 2 B, convert pass over q,k,v 2 B in / "
          f"1 B out per element; q/k abs-max from the centre's token sample): {ms:.3f} ms  {gb / ms:.2f} TB/s ({gb:.2f} GB)",
          flush=True)


if __name__ == "__main__":
    main()
