"""GPU parity of the fused q/k RMSNorm + rotary kernel (vorta_qk_norm_rope) against the oracle
and against the torch ops the reference runs at that point."""
import numpy as np
import pytest
import torch

from oracle import vorta_oracle as O
from _util import dev, rel_fro, rounded

pytestmark = pytest.mark.gpu

TOL = {torch.bfloat16: (3e-2, 5e-3), torch.float16: (4e-3, 8e-4)}  # max|d| (values up to ~6), rel. Frobenius


def _angles(S, seed):
    g = torch.Generator().manual_seed(seed)
    ang = torch.rand((S, 64), generator=g) * 6.28
    return ang.cos().repeat_interleave(2, dim=1).contiguous(), ang.sin().repeat_interleave(2, dim=1).contiguous()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_per_head_norm_rope_with_text_tail(dtype):
    """Hunyuan single-stream layout: video tokens are rotated, the text tail is only normalised (hunyuan.py:90-102);
    the tensor is the (B,S,H*D) projection output viewed as (H,S,D)."""
    from vorta_amd import ops
    H, S, T = 5, 333, 19
    g = torch.Generator().manual_seed(1)
    x = (torch.randn((S + T, H, 128), generator=g) * 2).to(dtype)
    w = (torch.rand(128, generator=g) + 0.5).to(dtype)
    cos, sin = _angles(S, 2)
    xd = x.to(dev())
    view = xd.permute(1, 0, 2)  # (H, S+T, D), stride_s = H*D
    ops.qk_norm_rope(view, w.to(dev()), 1e-6, cos=cos.to(dev()), sin=sin.to(dev()), rope_tokens=S)
    xr = rounded(x.float().numpy(), dtype).transpose(1, 0, 2)
    y = O.rms_norm(xr, rounded(w.float().numpy(), dtype), 1e-6)
    ref = np.concatenate([O.rope_interleaved(y[:, :S], cos.double().numpy(), sin.double().numpy()), y[:, S:]], axis=1)
    got = view.float().cpu().numpy()
    atol, rf = TOL[dtype]
    assert np.abs(got - ref).max() <= atol and rel_fro(got, ref) <= rf
    # the torch ops the reference runs (nn.RMSNorm in fp32 + the real-valued rotation)
    from vorta_amd.attention.hunyuan import apply_rotary_emb
    t = torch.nn.functional.rms_norm(x.to(dev()).permute(1, 0, 2).float(), (128,), w.to(dev()).float(), 1e-6)
    t = torch.cat([apply_rotary_emb(t[None, :, :S], (cos.to(dev()), sin.to(dev())))[0], t[:, S:]], dim=1)
    assert (view.float() - t).abs().max().item() <= atol


def test_offsets_and_no_weight():
    from vorta_amd import ops
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(3)
    x = torch.randn((3, 100, 128), generator=g).to(dtype).to(dev())
    orig = x.clone()
    cos, sin = _angles(40, 4)
    ops.qk_norm_rope(x, None, 1e-5, cos=cos.to(dev()), sin=sin.to(dev()), n_tokens=40, token_offset=30)
    assert torch.equal(x[:, :30], orig[:, :30]) and torch.equal(x[:, 70:], orig[:, 70:])
    ref = O.rope_interleaved(O.rms_norm(orig[:, 30:70].double().cpu().numpy(), None, 1e-5), cos.double().numpy(),
                             sin.double().numpy())
    assert np.abs(x[:, 30:70].float().cpu().numpy() - ref).max() <= TOL[dtype][0]


@pytest.mark.parametrize("heads", [12, 40, 24])
def test_across_heads_norm_complex_rope(heads):
    """Wan layout: RMSNorm over all H*D channels of a token (wan.py:85-89), then the complex rotation in float64
    (wan.py:34-37) -- compared with exactly those torch ops."""
    from vorta_amd import ops
    from vorta_amd.attention.wan import apply_rotary_emb
    dtype = torch.bfloat16
    S = 257
    g = torch.Generator().manual_seed(heads)
    x = torch.randn((1, S, heads * 128), generator=g).to(dtype).to(dev())
    w = (torch.rand(heads * 128, generator=g) + 0.5).to(dtype).to(dev())
    ang = torch.rand((S, 64), generator=g, dtype=torch.float64) * 6.28
    freqs = torch.polar(torch.ones_like(ang), ang).to(dev())[None, None]  # (1,1,S,D/2) complex128
    ref = torch.nn.functional.rms_norm(x.float(), (heads * 128,), w.float(), 1e-6).to(dtype)
    ref = apply_rotary_emb(ref.unflatten(2, (heads, -1)).transpose(1, 2), freqs)  # (1,H,S,D)
    cos = freqs.real[0, 0].repeat_interleave(2, dim=1).float().contiguous()
    sin = freqs.imag[0, 0].repeat_interleave(2, dim=1).float().contiguous()
    x_in = x[0].double().cpu().numpy()  # the kernel works in place
    view = x[0].unflatten(1, (heads, 128)).permute(1, 0, 2)  # (H,S,D): stride_h = D, stride_s = H*D
    ops.qk_norm_rope(view, w, 1e-6, cos=cos, sin=sin, across_heads=True)
    # the reference rounds to bf16 between norm and rotation; the fused kernel rounds once
    assert (view.float() - ref[0].float()).abs().max().item() <= 4e-2
    assert rel_fro(view.float().cpu().numpy(), ref[0].float().cpu().numpy()) <= 6e-3
    xo = O.rms_norm(x_in, w.double().cpu().numpy(), 1e-6)
    xo = O.rope_interleaved(xo.reshape(S, heads, 128).transpose(1, 0, 2), cos.double().cpu().numpy(), sin.double().cpu().numpy())
    assert np.abs(view.float().cpu().numpy() - xo).max() <= 3e-2


def test_bad_arguments():
    from vorta_amd import ops
    x = torch.zeros((2, 8, 64), dtype=torch.bfloat16, device=dev())
    with pytest.raises(Exception):
        ops.qk_norm_rope(x, None, 1e-6)  # head_dim 64
    x = torch.zeros((2, 8, 128), dtype=torch.bfloat16, device=dev())
    with pytest.raises(ValueError):
        ops.qk_norm_rope(x, None, 1e-6, cos=torch.zeros((4, 128), device=dev()), sin=torch.zeros((4, 128), device=dev()),
                         rope_tokens=8)


def test_full_size_bandwidth_smoke():
    """Hunyuan 129f shape: in-place on a (24, S+T, 128) view; prints the achieved HBM rate (2 x bytes / time)."""
    from vorta_amd import ops
    H, S, T = 24, 118800, 256
    x = torch.randn((S + T, H, 128), device=dev(), dtype=torch.bfloat16)
    w = torch.ones(128, device=dev(), dtype=torch.bfloat16)
    cos, sin = _angles(S, 9)
    cos, sin = cos.to(dev()), sin.to(dev())
    view = x.permute(1, 0, 2)
    ops.qk_norm_rope(view, w, 1e-6, cos=cos, sin=sin, rope_tokens=S)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.qk_norm_rope(view, w, 1e-6, cos=cos, sin=sin, rope_tokens=S)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    gbs = 2 * x.numel() * 2 / ms / 1e6
    print(f"\\nqk_norm_rope {x.numel() * 2 / 1e6:.0f} MB in place: {ms:.3f} ms = {gbs:.0f} GB/s (read+write)")
    assert torch.isfinite(view.float()).all()
    assert gbs > 300, gbs  # ~4 000 GB/s on a quiet box; the gate only catches a launch that is not HBM-streaming at all
