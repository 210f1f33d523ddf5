#!/usr/bin/env python
"""Bandwidth of the coreset selection (vorta_coreset_select) at the Hunyuan-129f geometry: 8 heads x 118 800 token rows of
256 B read once (243 MB), two int32 row lists written."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vorta_amd import ops


def main():
    dev = torch.device("cuda:0")
    latent, group, H, T = (33, 45, 80), (3, 3, 2), 24, 256
    S = latent[0] * latent[1] * latent[2]
    n_keep = 8  # rate 0.5 of the 17 margins of a (3,3,2) window
    for lay in ("head-major (H,S,D)", "token-major (S,H*D) view"):
        x = torch.randn((H, S + T, 128), device=dev, dtype=torch.float16) if lay.startswith("head") else \
            torch.randn((S + T, H, 128), device=dev, dtype=torch.float16).transpose(0, 1)
        hl = torch.arange(0, 8, dtype=torch.int32, device=dev)
        fn = lambda: ops.coreset_select(x, latent, group, n_keep, tail_first=S, n_tail=T, head_list=hl)
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        gb = 8 * S * 256 / 1e9
        print(f"vorta_coreset_select fp16, 8 of {H} heads x {S} rows, {lay}: {ms * 1e3:.1f} us  {gb / ms:.2f} TB/s ({gb * 1e3:.0f} MB read)", flush=True)


if __name__ == "__main__":
    main()
