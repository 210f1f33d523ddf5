#!/usr/bin/env python
"""Context for roofline.frac: what a plain library GEMM (hipBLASLt through torch.matmul) sustains on this box, i.e.
the practical MFMA ceiling under the chip's power/clock management, next to the 2.5 PFLOP/s nominal dense peak."""
import torch


def main():
    dev = torch.device("cuda:0")
    for dt in (torch.bfloat16, torch.float16):
        for n in (8192, 16384):
            a = torch.randn((n, n), device=dev).to(dt)
            b = torch.randn((n, n), device=dev).to(dt)
            for _ in range(3):
                a @ b
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            iters = 20 if n == 8192 else 6
            e0.record()
            for _ in range(iters):
                a @ b
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            print(f"torch.matmul {dt} {n}^3: {ms:.3f} ms  {2.0 * n ** 3 / ms / 1e9:.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
