#!/usr/bin/env python
"""Feasibility check: how much does a per-channel bias in K (common-mode component of the keys: softmax-invariant) cost
the e4m3 path, and how much of it does subtracting the per-head key mean before the conversion give back?"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from vorta_amd import ops


def psnr(x, ref):
    x, ref = x.float(), ref.float()
    mse = torch.mean((x - ref) ** 2).item()
    rng = (ref.max() - ref.min()).item()
    return 10 * math.log10(rng * rng / max(mse, 1e-30)), math.sqrt(mse / torch.mean(ref ** 2).item())


def main():
    dev = torch.device("cuda:0")
    H, S = 2, 16384
    g = torch.Generator(device=dev).manual_seed(0)
    for bias in (0.0, 1.0, 3.0, 8.0):
        for qscale in (1.0, 3.0):
            q = (torch.randn((H, S, 128), generator=g, device=dev) * qscale).to(torch.bfloat16)
            kb = torch.randn((H, 1, 128), generator=g, device=dev) * bias
            k = (torch.randn((H, S, 128), generator=g, device=dev) + kb).to(torch.bfloat16)
            v = torch.randn((H, S, 128), generator=g, device=dev).to(torch.bfloat16)
            ref = torch.empty_like(q)
            ops.attn_fwd(q, k, v, ref, n_q=S, n_kv=S)
            outs = {}
            for name, kk in (("as is", k), ("centred", (k.float() - k.float().mean(1, keepdim=True)).to(torch.bfloat16))):
                f8 = ops.fp8_quantize_qkv(q, kk, v)
                o = torch.empty_like(q)
                ops.attn_fwd(f8.q, f8.k, f8.v, o, n_q=S, n_kv=S, v_descale=f8.v_descale)
                outs[name] = psnr(o, ref)
            print(f"K bias {bias:3.1f} sigma, q scale {qscale}: " +
                  "; ".join(f"{n}: PSNR {p:5.1f} dB rel {r:.4f}" for n, (p, r) in outs.items()), flush=True)


if __name__ == "__main__":
    main()
