"""GPU: the geometries of BASELINE.json's configs at full size, checked through size-independent properties and
sampled rows against the oracle (the oracle cannot hold an S x S score matrix at these sizes).

  configs[0]  Wan-2.1 1.3B 49x320x512  (13,20,32)  S =   8 320  native (dense) attention, 12 heads
  configs[1]  Wan-2.1 1.3B 81x480x832  (21,30,52)  S =  32 760  routed, tile (7,6,4), coreset (3,3,2)
  configs[2]  HunyuanVideo 129x720x1280 (33,45,80) S = 118 800  routed: tests/test_hip_experts.py::test_full_size_*
  configs[4]  Wan-2.1 14B 81x720x1280  (21,45,80)  S =  75 600  routed, tile (7,9,8), coreset (3,3,2), 40 heads (bf16;
              the fp8 MFMA path of that config is not built)
"""
import numpy as np
import pytest
import torch

from oracle import vorta_oracle as O
from _util import ATOL_SAME, check, dev

pytestmark = pytest.mark.gpu
WINDOW = (3, 3, 3)


def _rand(shape, seed, dtype):
    gen = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=gen).to(dtype).to(dev())


def test_config0_wan13b_49f_native_attention():
    from vorta_amd.patch import wan_pixel2token
    from vorta_amd.routed import dense_attention
    latent = wan_pixel2token((49, 320, 512))
    assert latent == (13, 20, 32)
    S, H, dtype = 13 * 20 * 32, 12, torch.bfloat16
    q, k, v = (_rand((1, H, S, 128), s, dtype) for s in (1, 2, 3))
    out = dense_attention(q, k, v)
    for h in (0, 7):  # two full heads against the oracle
        ref = O.dense_attention(q[0, h].double().cpu().numpy(), k[0, h].double().cpu().numpy(), v[0, h].double().cpu().numpy())
        check(out[0, h], ref, dtype)
        # the north star's image-level bar, applied to the operator: PSNR >= 40 dB against the float64 result
        got = out[0, h].double().cpu().numpy()
        psnr = 20.0 * np.log10(np.abs(ref).max() / np.sqrt(np.mean((got - ref) ** 2)))
        assert psnr >= 40.0, psnr
    # every head: softmax rows sum to one (constant V reproduced)
    const = _rand((1, H, 1, 128), 4, dtype)
    out2 = dense_attention(q, k, const.expand(1, H, S, 128).contiguous())
    assert (out2.float() - const.float()).abs().max().item() <= 2e-2


def _sampled_expert_checks(model, latent, tile, group, H, experts, dtype, seed, text=(0, 0)):
    """Routed op at full size; per expert compare sampled query rows with the oracle run on the key/query lists the
    kernels used (tables are bit-exact vs the reference at small size), plus the structural properties."""
    from vorta_amd import ops
    from vorta_amd.routed import HeadRouting, RoutedGeometry, routed_attention
    S = latent[0] * latent[1] * latent[2]
    T, te = text
    q, k, v = (_rand((1, H, S + T, 128), seed + i, dtype) for i in range(3))
    geom = RoutedGeometry(latent, tile, WINDOW, group, 0.5, dev())
    out = routed_attention(q, k, v, HeadRouting.from_expert_ids(experts, dev()), geom, model=model, text_len=T,
                           text_valid=te)
    gen = torch.Generator().manual_seed(seed)
    tol = ATOL_SAME[dtype]
    f64 = lambda t: t.double().cpu().numpy()
    h_full, h_low, h_sl = (experts.index(e) for e in (0, 1, 2))
    # full expert: sampled rows vs dense oracle
    rows = torch.randint(0, S, (32,), generator=gen)
    ref = O.dense_attention(f64(q[0, h_full, rows]), f64(k[0, h_full, :S + te]), f64(v[0, h_full, :S + te]))
    assert np.abs(out[0, h_full, rows.to(dev())].float().cpu().numpy() - ref).max() <= tol
    # coreset expert: rows of the packed sequence vs dense oracle over the kept keys; dropped margins == their centre
    keep_q, drop_q = ops.coreset_select(q[0, h_low:h_low + 1], latent, group, geom.n_keep, tail_first=S, n_tail=T)
    keep_k = keep_q if model == "wan" else ops.coreset_select(k[0, h_low:h_low + 1], latent, group, geom.n_keep,
                                                              tail_first=S, n_tail=te, want_drop=False)[0]
    kk = keep_k[0, :geom.S_low + te].long()
    pos = torch.randint(0, geom.S_low, (32,), generator=gen)
    qr = keep_q[0].cpu()[pos].long()
    ref = O.dense_attention(f64(q[0, h_low, qr]), f64(k[0, h_low, kk]), f64(v[0, h_low, kk]))
    assert np.abs(out[0, h_low, qr.to(dev())].float().cpu().numpy() - ref).max() <= tol
    centres = keep_q[0, :geom.G].long()
    assert torch.equal(out[0, h_low][drop_q[0].long()], out[0, h_low][centres][:, None].expand(-1, drop_q.shape[-1], -1))
    # sliding expert: sampled rows vs dense oracle over the tile's key list
    q_rows, kv_rows, _ = geom.sta_tables(te)
    pos = torch.randint(0, S, (24,), generator=gen)
    for p in pos.tolist():
        keys = kv_rows[p // geom.tok].long()
        r = int(q_rows[p])
        ref = O.dense_attention(f64(q[0, h_sl, r:r + 1]), f64(k[0, h_sl, keys]), f64(v[0, h_sl, keys]))
        assert np.abs(out[0, h_sl, r].float().cpu().numpy() - ref[0]).max() <= tol
    return out


def test_config1_wan13b_81f_routed():
    from vorta_amd.patch import wan_pixel2token
    assert wan_pixel2token((81, 480, 832)) == (21, 30, 52)
    experts = [0, 1, 2, 2, 1, 0, 1, 2, 0, 0, 2, 1]
    _sampled_expert_checks("wan", (21, 30, 52), (7, 6, 4), (3, 3, 2), 12, experts, torch.bfloat16, seed=100)


def test_config4_wan14b_81f_geometry_routed_bf16():
    from vorta_amd.patch import wan_pixel2token
    assert wan_pixel2token((81, 720, 1280)) == (21, 45, 80)
    experts = [0, 1, 2, 1]  # 4 of the 40 heads are enough to exercise the geometry at S = 75 600
    _sampled_expert_checks("wan", (21, 45, 80), (7, 9, 8), (3, 3, 2), 4, experts, torch.bfloat16, seed=200)


def test_reference_native_geometry_hunyuan_117f():
    """The authors' own benchmark shape (vorta/constants.py:8-12; tile (6,9,8), coreset (2,3,2))."""
    experts = [0, 1, 2]
    _sampled_expert_checks("hunyuan", (30, 45, 80), (6, 9, 8), (2, 3, 2), 3, experts, torch.float16, seed=300,
                           text=(256, 77))
