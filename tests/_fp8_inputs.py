"""Input families for the accuracy gates of the e4m3 path (tests/test_hip_fp8.py, tools/dbg/fp8_structured.py): what q, k, v
of a trained video DiT can look like beyond white noise -- a common component in q and in k, heavy tails, fields that are
smooth over the latent grid, a peaked softmax, outlier channels from qk-norm weights."""
import math

import torch


def psnr(x, ref):
    """(PSNR over the data range max - min, PSNR over max |ref|, relative rms error)"""
    x, ref = x.float(), ref.float()
    mse = torch.mean((x - ref) ** 2).item()
    rng, peak = (ref.max() - ref.min()).item(), ref.abs().max().item()
    f = lambda r: 10.0 * math.log10(r * r / max(mse, 1e-30))
    return f(rng), f(peak), math.sqrt(mse / torch.mean(ref ** 2).item())


def robust_psnr(x, ref, q=0.999):
    """PSNR over a ROBUST peak, the q-quantile of |ref| (a sample of <= 4M elements): on heavy-tailed inputs max |ref| sits
    far above the bulk and a PSNR over it says little about the bulk's error"""
    x, ref = x.float(), ref.float()
    a = ref.abs().flatten()
    if a.numel() > (1 << 22):
        a = a[:: a.numel() // (1 << 22)]
    peak = torch.quantile(a, q).item()
    mse = torch.mean((x - ref) ** 2).item()
    return 10.0 * math.log10(peak * peak / max(mse, 1e-30))


def unit_rms(x):
    return x / x.pow(2).mean(-1, keepdim=True).sqrt()


def smoothed(q, k):
    """per-channel smoothing q diag(s), k diag(1/s) with s = sqrt(amax_k / amax_q) (SmoothQuant-style): exact for the
    scores; helps a FIXED-point format, does nothing for a floating-point one (the relative rounding error of every
    product is what it was)"""
    aq, ak = q.abs().amax(1, keepdim=True).float(), k.abs().amax(1, keepdim=True).float()
    s = (ak / aq).sqrt()
    return q.float() * s, k.float() / s


def families(latent, H, T, gen, dev):
    """yields (name, q, k, v) float32 (H, S + T, 128)"""
    S = latent[0] * latent[1] * latent[2]
    N = S + T
    randn = lambda *shape: torch.randn(shape, generator=gen, device=dev)

    def smooth_field(scale=4):
        """field correlated over the latent grid: coarse noise upsampled trilinearly + 30 % white noise"""
        t, h, w = latent
        c = randn(H * 128, 1, max(2, t // scale), max(2, h // scale), max(2, w // scale))
        f = torch.nn.functional.interpolate(c, size=(t, h, w), mode="trilinear", align_corners=False)
        f = f.reshape(H, 128, S).transpose(1, 2)
        f = f / f.std() + 0.3 * randn(H, S, 128)
        return torch.cat([f, randn(H, T, 128)], 1) if T else f

    w = torch.ones(128, device=dev)
    w[torch.randperm(128, generator=gen, device=dev)[:6]] = torch.tensor([10., 15., 20., 20., 25., 30.], device=dev)
    mu_q, mu_k = unit_rms(randn(H, 1, 128)), unit_rms(randn(H, 1, 128))

    def t3():  # Student-t with 3 degrees of freedom, unit variance: normal / sqrt(chi2_3 / 3) / sqrt(3)
        z = randn(H, N, 128)
        c = randn(H, N, 128, 3).pow(2).sum(-1) / 3.0
        return z / c.sqrt() / math.sqrt(3.0)

    yield "white", randn(H, N, 128), randn(H, N, 128), randn(H, N, 128)
    yield "common3", randn(H, N, 128) + 3.0 * mu_q, randn(H, N, 128) + 3.0 * mu_k, randn(H, N, 128)
    yield "student_t3", t3(), t3(), t3()
    yield "smooth", smooth_field(), smooth_field(), smooth_field()
    yield "peaked", 2 * smooth_field(), 2 * smooth_field(), randn(H, N, 128)
    # qk-RMSNorm with outlier weights (10-30x) on the same 6 channels of q and k; the logits kept O(1-10) like a trained
    # model's, whatever its norm weights
    yield "outlier_w", unit_rms(randn(H, N, 128)) * w / 4.5, unit_rms(randn(H, N, 128)) * w / 4.5, randn(H, N, 128)
    yield "outlier_w_common", unit_rms(0.25 * randn(H, N, 128) + 2.0 * mu_q) * w, \
        unit_rms(0.25 * randn(H, N, 128) + 2.0 * mu_k) * w, randn(H, N, 128)


NAMES = {"white": "white noise", "common3": "common component of 3 sigma in q and in k", "student_t3": "Student-t(3) q, k, v",
         "smooth": "q, k, v smooth over the latent grid", "peaked": "smooth q, k x 2 (peaked softmax), white v",
         "outlier_w": "qk-norm weights 10-30x on 6 channels, logits O(1-10)",
         "outlier_w_common": "the same with a common part of 2 sigma in q and k"}
