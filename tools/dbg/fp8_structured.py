#!/usr/bin/env python
"""fp8 (e4m3) routed attention against the 16-bit kernels on inputs that are NOT white noise (tests/_fp8_inputs.py):
prints, per input family and expert, PSNR over the data range, PSNR over max|x| and the relative rms error at Wan-14B-81f or
Hunyuan-129f geometry, and for the outlier-channel families the same after per-channel smoothing of q and k."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from _fp8_inputs import NAMES, families, psnr, smoothed

from vorta_amd.routed import HeadRouting, RoutedGeometry, routed_attention

ap = argparse.ArgumentParser()
ap.add_argument("--geometry", default="wan14b-81f", choices=["wan14b-81f", "hunyuan-129f"])
ap.add_argument("--dtype", default="bf16")
args = ap.parse_args()
dev = torch.device("cuda:0")
dt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
if args.geometry == "wan14b-81f":
    latent, tile, window, group, model, T, te = (21, 45, 80), (7, 9, 8), (3, 3, 3), (3, 3, 2), "wan", 0, 0
else:
    latent, tile, window, group, model, T, te = (33, 45, 80), (11, 9, 8), (3, 3, 3), (3, 3, 2), "hunyuan", 256, 96
S = latent[0] * latent[1] * latent[2]
H = 3
geom = RoutedGeometry(latent, tile, window, group, 0.5, dev)
routing = HeadRouting.from_expert_ids([0, 1, 2], dev)
gen = torch.Generator(device=dev).manual_seed(1234)
names = ["full", "coreset", "sliding"]
kw = dict(model=model, text_len=T, text_valid=te)
print(f"{args.geometry} {args.dtype}: PSNR over data range dB | PSNR over max|x| dB | rel rms")
for key, q, k, v in families(latent, H, T, gen, dev):
    q16, k16, v16 = (x.to(dt)[None].contiguous() for x in (q, k, v))
    ref = routed_attention(q16, k16, v16, routing, geom, **kw)
    out = routed_attention(q16, k16, v16, routing, geom, fp8=True, **kw)
    torch.cuda.synchronize()
    row = ["%s %5.1f | %5.1f | %.4f" % ((names[h],) + psnr(out[0, h, :S + te], ref[0, h, :S + te])) for h in range(H)]
    print(f"{NAMES[key]:62s} " + "   ".join(row), flush=True)
    if key.startswith("outlier"):
        qs, ks = (x.to(dt)[None].contiguous() for x in smoothed(q, k))
        ref2 = routed_attention(qs, ks, v16, routing, geom, **kw)
        out2 = routed_attention(qs, ks, v16, routing, geom, fp8=True, **kw)
        torch.cuda.synchronize()
        # (the coreset expert ranks by cosine similarity, which smoothing changes: not comparable)
        row = ["%s %5.1f | %5.1f | %.4f" % ((names[h],) + psnr(out2[0, h, :S + te], ref2[0, h, :S + te])) for h in (0, 2)]
        print(f"{'    after q diag(s), k diag(1/s) per channel':62s} " + "   ".join(row), flush=True)
