#!/usr/bin/env python
"""Feasibility check: does a per-channel mean in V cost the e4m3 path anything (out = sum p v = sum p (v - c) + c)?"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from vorta_amd import ops


def err(x, ref):
    x, ref = x.float(), ref.float()
    mse = torch.mean((x - ref) ** 2).item()
    rng = (ref.max() - ref.min()).item()
    dev = ref - ref.mean(1, keepdim=True)  # what differs between queries
    return 10 * math.log10(rng * rng / max(mse, 1e-30)), math.sqrt(mse / torch.mean(dev ** 2).item())


def main():
    dev = torch.device("cuda:0")
    H, S = 2, 16384
    g = torch.Generator(device=dev).manual_seed(0)
    for bias in (0.0, 1.0, 3.0, 8.0):
        for qscale in (1.0, 3.0):
            q = (torch.randn((H, S, 128), generator=g, device=dev) * qscale).to(torch.bfloat16)
            k = torch.randn((H, S, 128), generator=g, device=dev).to(torch.bfloat16)
            vb = torch.randn((H, 1, 128), generator=g, device=dev) * bias
            v = (torch.randn((H, S, 128), generator=g, device=dev) + vb).to(torch.bfloat16)
            ref = torch.empty_like(q)
            ops.attn_fwd(q, k, v, ref, n_q=S, n_kv=S)
            outs = {}
            c = v.float().mean(1, keepdim=True)
            for name, vv, add in (("as is", v, 0.0), ("centred", (v.float() - c).to(torch.bfloat16), c)):
                f8 = ops.fp8_quantize_qkv(q, k, vv, center_k=True)
                o = torch.empty_like(q)
                ops.attn_fwd(f8.q, f8.k, f8.v, o, n_q=S, n_kv=S, v_descale=f8.v_descale)
                outs[name] = err((o.float() + add).to(torch.bfloat16), ref)
            print(f"V bias {bias:3.1f} sigma, q scale {qscale}: " +
                  "; ".join(f"{n}: PSNR {p:5.1f} dB, error / between-query variation {r:.4f}" for n, (p, r) in outs.items()),
                  flush=True)


if __name__ == "__main__":
    main()
