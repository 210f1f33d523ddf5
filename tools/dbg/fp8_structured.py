#!/usr/bin/env python
"""fp8 (e4m3) routed attention against the 16-bit kernels on inputs that are NOT white noise: outlier channels from
norm weights, common components in q and k, heavy tails, spatially smooth fields.  Prints, per input family and expert,
PSNR over the data range, PSNR over max|x| and the relative rms error (Wan-14B-81f or Hunyuan-129f geometry)."""
import argparse
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

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
N = S + T
H = 3
geom = RoutedGeometry(latent, tile, window, group, 0.5, dev)
routing = HeadRouting.from_expert_ids([0, 1, 2], dev)
gen = torch.Generator(device=dev).manual_seed(1234)


def randn(*shape):
    return torch.randn(shape, generator=gen, device=dev)


def unit_rms(x):
    return x / x.pow(2).mean(-1, keepdim=True).sqrt()


def smooth_field(scale=4):
    """(H, N, 128) field correlated over the latent grid: coarse noise upsampled trilinearly + 30 % white noise"""
    t, h, w = latent
    c = randn(H * 128, 1, max(2, t // scale), max(2, h // scale), max(2, w // scale))
    f = torch.nn.functional.interpolate(c, size=(t, h, w), mode="trilinear", align_corners=False)
    f = f.reshape(H, 128, S).transpose(1, 2)
    f = f / f.std() + 0.3 * randn(H, S, 128)
    if T:
        f = torch.cat([f, randn(H, T, 128)], 1)
    return f


def families():
    w = torch.ones(128, device=dev)
    w[torch.randperm(128, generator=gen, device=dev)[:6]] = torch.tensor([10., 15., 20., 20., 25., 30.], device=dev)
    mu_q, mu_k = unit_rms(randn(H, 1, 128)), unit_rms(randn(H, 1, 128))
    t3 = lambda: torch.distributions.StudentT(3.0).sample((H, N, 128)).to(dev) / math.sqrt(3.0)
    yield "white noise", randn(H, N, 128), randn(H, N, 128), randn(H, N, 128)
    # qk-RMSNorm with outlier weights on the same 6 channels of q and k; softmax temperature kept sane by scaling the
    # NON-common part down (a trained model's logits are O(1-10), whatever its norm weights)
    yield "outlier norm weights (10-30x on 6 channels), common part 2 sigma", \
        unit_rms(0.25 * randn(H, N, 128) + 2.0 * mu_q) * w, unit_rms(0.25 * randn(H, N, 128) + 2.0 * mu_k) * w, randn(H, N, 128)
    yield "outlier norm weights, no common part, logits / 20", unit_rms(randn(H, N, 128)) * w / 4.5, \
        unit_rms(randn(H, N, 128)) * w / 4.5, randn(H, N, 128)
    yield "common component 3 sigma in q and in k", randn(H, N, 128) + 3.0 * mu_q, randn(H, N, 128) + 3.0 * mu_k, randn(H, N, 128)
    yield "Student-t(3) q, k, v", t3(), t3(), t3()
    yield "smooth fields (q, k, v correlated over the latent grid)", smooth_field(), smooth_field(), smooth_field()
    yield "smooth q, k (x 2: peaked softmax), white v", 2 * smooth_field(), 2 * smooth_field(), randn(H, N, 128)


def psnr(x, ref):
    x, ref = x.float(), ref.float()
    mse = torch.mean((x - ref) ** 2).item()
    rng, peak = (ref.max() - ref.min()).item(), ref.abs().max().item()
    f = lambda r: 10.0 * math.log10(r * r / max(mse, 1e-30))
    return f(rng), f(peak), math.sqrt(mse / torch.mean(ref ** 2).item())


def smoothed(q, k):
    """per-channel smoothing q diag(s), k diag(1/s) with s = sqrt(amax_k / amax_q) (SmoothQuant-style): exact for the
    scores; would help a FIXED-point format, and is expected to do nothing for a floating-point one (the relative
    rounding error of every product is the same before and after)"""
    aq, ak = q.abs().amax(1, keepdim=True).float(), k.abs().amax(1, keepdim=True).float()
    s = (ak / aq).sqrt()
    return (q.float() * s), (k.float() / s)


names = ["full", "coreset", "sliding"]
print(f"{args.geometry} {args.dtype}: PSNR over data range dB | PSNR over max|x| dB | rel rms")
for name, q, k, v in families():
    q, k, v = (x.to(dt)[None].contiguous() for x in (q, k, v))
    kw = dict(model=model, text_len=T, text_valid=te)
    ref = routed_attention(q, k, v, routing, geom, **kw)
    out = routed_attention(q, k, v, routing, geom, fp8=True, **kw)
    torch.cuda.synchronize()
    row = []
    for h in range(H):
        a, b, c = psnr(out[0, h, :S + te], ref[0, h, :S + te])
        row.append(f"{names[h]} {a:5.1f} | {b:5.1f} | {c:.4f}")
    print(f"{name:70s} " + "   ".join(row), flush=True)
    if "outlier" in name:
        qs, ks = smoothed(q[0], k[0])
        out = routed_attention(qs.to(dt)[None], ks.to(dt)[None], v, routing, geom, fp8=True, **kw)
        ref2 = routed_attention(qs.to(dt)[None], ks.to(dt)[None], v, routing, geom, **kw)
        torch.cuda.synchronize()
        row = []
        for h in (0, 2):  # (the coreset expert ranks by cosine similarity, which smoothing changes: not comparable)
            a, b, c = psnr(out[0, h, :S + te], ref2[0, h, :S + te])
            row.append(f"{names[h]} {a:5.1f} | {b:5.1f} | {c:.4f}")
        print(f"{'    the same with q diag(s), k diag(1/s) per channel':70s} " + "   ".join(row), flush=True)
    del q, k, v, ref, out
