"""Synthetic post-RoPE q, k, v WITH spatio-temporal structure, for the path-level accuracy of the ROUTED operator against
native (all-dense) attention (north_star: PSNR >= 40 dB vs --native_attention).

White noise is the floor of the method (22 dB, tests/test_hip_experts.py::test_routed_vs_native_attention_operator_psnr):
no expert has anything to exploit.  VORTA's premise (/root/reference README, hunyuan.py:562-605) is that a trained video DiT's
heads are either LOCAL -- attention mass sits in a spatio-temporal neighbourhood, which the sliding-tile expert keeps -- or
REDUNDANT over neighbouring tokens -- which the coreset expert pools -- and that the router learns which head is which.  These
generators build heads of exactly those two kinds with a knob for how much unstructured noise rides on top, so the operator
can be measured where the method is supposed to work, and where it stops working:

  * `local_head`: q_i = k_i-like random Fourier features of the token's (t, h, w) position: q_i . k_j / sqrt(D) =
    amp * exp(-|(p_i - p_j) / ell|^2 / 2) + noise terms -- a Gaussian neighbourhood of `ell` tokens per dimension, `amp` logits
    above the background; v a smooth field;
  * `redundant_head`: q, k, v constant over blocks of the coreset window's size -- the tokens of a window are duplicates, the
    limit of "neighbouring tokens carry the same information";
  * `noise` in [0, 1]: the share of white noise mixed into every tensor (1 = the white-noise floor).
"""
import math

import torch


def _positions(latent, dev):
    t, h, w = latent
    g = torch.stack(torch.meshgrid(torch.arange(t, device=dev), torch.arange(h, device=dev), torch.arange(w, device=dev),
                                   indexing="ij"), -1)
    return g.reshape(-1, 3).float()


def _smooth(latent, channels, scale, gen, dev):
    t, h, w = latent
    c = torch.randn((channels, 1, max(2, t // scale), max(2, h // scale), max(2, w // scale)), generator=gen, device=dev)
    f = torch.nn.functional.interpolate(c, size=(t, h, w), mode="trilinear", align_corners=False)
    f = f.reshape(channels, t * h * w).t()
    return f / f.std()


def local_head(latent, ell, amp, noise, gen, dev, D=128):
    """(q, k, v) float32 (S, D) of one LOCAL head"""
    pos = _positions(latent, dev)
    S = pos.shape[0]
    W = torch.randn((3, D // 2), generator=gen, device=dev) / torch.tensor(ell, device=dev, dtype=torch.float32).view(3, 1)
    b = torch.rand((D // 2,), generator=gen, device=dev) * (2 * math.pi)
    ang = pos @ W + b
    phi = torch.cat([ang.cos(), ang.sin()], 1) / math.sqrt(D // 2)  # |phi| = 1, phi_i . phi_j ~ exp(-|dp / ell|^2 / 2)
    a = math.sqrt(amp * math.sqrt(D))
    mix = lambda x: math.sqrt(1.0 - noise ** 2) * x + noise * torch.randn(x.shape, generator=gen, device=dev)
    q = mix(a * phi)
    k = mix(a * phi)
    v = mix(_smooth(latent, D, 3, gen, dev))
    return q, k, v


def _blocks(latent, channels, block, gen, dev):
    """field that is CONSTANT over blocks of `block` tokens per dimension (the blocks tile the latent grid), unit variance"""
    t, h, w = latent
    n = [-(-a // b) for a, b in zip(latent, block)]
    c = torch.randn((channels, n[0], n[1], n[2]), generator=gen, device=dev)
    f = c.repeat_interleave(block[0], 1).repeat_interleave(block[1], 2).repeat_interleave(block[2], 3)[:, :t, :h, :w]
    return f.reshape(channels, t * h * w).t().contiguous()


def redundant_head(latent, block, noise, gen, dev, D=128, gain=2.0):
    """(q, k, v) float32 (S, D) of one REDUNDANT head: q, k, v constant over blocks of `block` tokens (a multiple of the coreset
    window: the tokens of a window are duplicates, what the coreset expert assumes), white noise of share `noise` on top;
    `gain`: logit spread, so the softmax is not flat"""
    mix = lambda x: math.sqrt(1.0 - noise ** 2) * x + noise * torch.randn(x.shape, generator=gen, device=dev)
    q = mix(_blocks(latent, D, block, gen, dev)) * math.sqrt(gain)
    k = mix(_blocks(latent, D, block, gen, dev)) * math.sqrt(gain)
    v = mix(_blocks(latent, D, block, gen, dev))
    return q, k, v


def structured_layer(latent, experts, tile, group, noise, gen, dev, D=128, amp=16.0, matched=True):
    """(q, k, v) float32 (1, H, S, D): every head gets the structure its expert exploits (`matched`: what a trained router
    does -- sliding-tile heads local at a third of a tile, coreset heads redundant, full-attention heads anything) or the
    other kind (the control: a local head pooled, a redundant head windowed)."""
    qs, ks, vs = [], [], []
    ell = tuple(max(x / 3.0, 0.75) for x in tile)
    for e in experts:
        kind = {0: "white", 1: "redundant", 2: "local"}[int(e)]
        if not matched and kind != "white":
            kind = "local" if kind == "redundant" else "redundant"
        if kind == "white":
            q, k, v = (torch.randn((latent[0] * latent[1] * latent[2], D), generator=gen, device=dev) for _ in range(3))
        elif kind == "local":
            q, k, v = local_head(latent, ell, amp, noise, gen, dev, D)
        else:
            q, k, v = redundant_head(latent, group, noise, gen, dev, D)
        qs.append(q); ks.append(k); vs.append(v)
    return tuple(torch.stack(x)[None] for x in (qs, ks, vs))
