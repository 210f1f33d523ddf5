#!/usr/bin/env python
"""Quick kernel timing on the GPU box: dense attention TFLOP/s of vorta_attn_fwd vs torch SDPA."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vorta_amd import ops


def timeit(fn, iters):
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--S", type=int, default=32760)
    ap.add_argument("--H", type=int, default=12)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--block_rows", type=int, default=0)
    ap.add_argument("--sdpa", action="store_true")
    ap.add_argument("--variants", default="1,2")  # 1 plain, 2 pipelined
    ap.add_argument("--rounds", type=int, default=1)
    ap.add_argument("--fp8", default="", help="comma list of vorta_attn_fp8_ext flag values to time the e4m3 kernel with (e.g. 0,1)")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float16
    dev = torch.device("cuda:0")
    q, k, v = (torch.randn((a.H, a.S, 128), device=dev, dtype=dt) for _ in range(3))
    o = torch.empty_like(q)
    flops = 4.0 * a.S * a.S * 128 * a.H
    for rnd in range(a.rounds):
      for remap in (0,):
        ops.NO_XCD_REMAP = remap
        for br in ([a.block_rows] if a.block_rows else [256, 128]):
            for var in [int(x) for x in a.variants.split(",")]:
                ms = timeit(lambda: ops.attn_fwd(q, k, v, o, n_q=a.S, n_kv=a.S, block_rows=br, variant=var), a.iters)
                print(f"vorta_attn_fwd S={a.S} H={a.H} {a.dtype} block_rows={br} variant={var} no_xcd_remap={remap}: "
                      f"{ms:.3f} ms  {flops/ms/1e9:.1f} TFLOP/s", flush=True)
    if a.fp8:
        f8 = ops.fp8_quantize_qkv(q, k, v)
        for rnd in range(a.rounds):
            for br in ([a.block_rows] if a.block_rows else [256, 128]):
                for fl in [int(x) for x in a.fp8.split(",")]:
                    ms = timeit(lambda: ops.attn_fwd(f8.q, f8.k, f8.v, o, n_q=a.S, n_kv=a.S, block_rows=br,
                                                     v_descale=f8.v_descale, fp8_opts=dict(flags=fl)), a.iters)
                    print(f"vorta_attn_fwd_fp8 S={a.S} H={a.H} out {a.dtype} block_rows={br} flags={fl}: "
                          f"{ms:.3f} ms  {flops/ms/1e9:.1f} TFLOP/s", flush=True)
        ms = timeit(lambda: ops.fp8_quantize_qkv(q, k, v, out=f8), a.iters)
        gb = 3 * q.numel() * 5 / 1e9
        print(f"vorta_fp8_quantize_qkv: {ms:.3f} ms  {gb/ms:.2f} TB/s (5 B per element)", flush=True)
    if a.sdpa:
        import torch.nn.functional as F
        q4, k4, v4 = q[None], k[None], v[None]
        ms = timeit(lambda: F.scaled_dot_product_attention(q4, k4, v4), a.iters)
        print(f"torch SDPA: {ms:.3f} ms  {flops/ms/1e9:.1f} TFLOP/s")
        ref = F.scaled_dot_product_attention(q4, k4, v4)[0]
        print("max|vorta - sdpa| =", (o.float() - ref.float()).abs().max().item())


if __name__ == "__main__":
    main()
