"""GPU, and only where `diffusers` is installed (it is not in the authoring image: these tests SKIP there and fire on any box
that has it): the patches against the REAL classes the reference patches -- `HunyuanVideoTransformer3DModel`
(/root/reference/vorta/patch/modeling_hunyuan.py:648-723) and `WanTransformer3DModel` (modeling_wan.py:265-324) -- built from
tiny configurations (random weights, head_dim 128, 2 + 2 / 2 blocks).  Every other patch / pipeline test runs on the structural
stand-ins of tests/_mini_diffusers.py; this file is what would notice a diffusers attribute or call-protocol drift.

Checked: every attribute name `vorta_amd/patch/_engine.py` / `modeling_*.py` rely on exists; `apply_vorta_transformer` with
every router forced to the dense expert gives the forward of `apply_sp_flashattn_transformer` on the same weights (both run the
HIP dense kernel) and stays within 16-bit tolerance of the STOCK diffusers processors (torch SDPA); a routed forward with
spread routers runs, is finite and returns the reference's 5-tuple with one score tensor per block."""
import pytest
import torch

pytestmark = pytest.mark.gpu
diffusers = pytest.importorskip("diffusers", reason="diffusers is not installed in this image: the stand-ins of "
                                                   "tests/_mini_diffusers.py cover the patches; this file fires where it is")

LATENT, TILE, WINDOW, GROUP = (4, 6, 8), (2, 3, 4), (3, 3, 3), (2, 3, 2)


def dev():
    return torch.device("cuda", 0)


def _build(cls, **config):
    try:
        return cls(**config)
    except TypeError as exc:  # another diffusers version names its constructor arguments differently
        pytest.skip(f"{cls.__name__} of diffusers {diffusers.__version__} does not take this tiny configuration: {exc}")


def _all_dense_routers(model):
    for m in model.modules():
        if type(m).__name__ == "Router":
            with torch.no_grad():
                m.linear.weight.zero_()
                b = torch.zeros(m.heads, m.num_experts)
                b[:, 0] = 8.0
                m.linear.bias.copy_(b.reshape(-1).to(m.linear.bias))


def _spread_routers(model, seed):
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if type(m).__name__ == "Router":
            m.linear.weight.data = (torch.randn(m.linear.weight.shape, generator=g) * 0.5).to(m.linear.weight)
            m.linear.bias.data = (torch.randn(m.linear.bias.shape, generator=g) * 2.0).to(m.linear.bias)


def _close(a, b, tol):
    a, b = a.float(), b.float()
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    err, mag = float((a - b).abs().max()), float(b.abs().max())
    assert err <= tol * max(mag, 1.0), (err, mag)


def test_hunyuan_real_class_attribute_names_and_all_dense_equivalence():
    from diffusers import HunyuanVideoTransformer3DModel

    from vorta.patch.modeling_hunyuan import apply_sp_flashattn_transformer, apply_vorta_transformer
    from vorta.patch.utils import prepare_hunyuan_self_attn_kwargs
    torch.manual_seed(0)
    model = _build(HunyuanVideoTransformer3DModel, in_channels=4, out_channels=4, num_attention_heads=2, attention_head_dim=128,
                   num_layers=2, num_single_layers=2, num_refiner_layers=1, mlp_ratio=2.0, patch_size=2, patch_size_t=1,
                   qk_norm="rms_norm", guidance_embeds=True, text_embed_dim=32, pooled_projection_dim=16, rope_theta=256.0,
                   rope_axes_dim=(16, 56, 56)).to(dev()).to(torch.bfloat16).eval()
    # the attribute names the patches rely on (modeling_hunyuan.py:660-701 of the reference reads the same ones)
    assert len(model.transformer_blocks) == 2 and len(model.single_transformer_blocks) == 2
    assert hasattr(model, "rope") and hasattr(model, "norm_out") and hasattr(model.time_text_embed, "timestep_embedder")
    for block in list(model.transformer_blocks) + list(model.single_transformer_blocks):
        attn = block.attn
        assert attn.heads == 2 and callable(attn.set_processor)
        for name in ("to_q", "to_k", "to_v", "norm_q", "norm_k"):
            assert getattr(attn, name) is not None, name
        norm = block.norm1 if hasattr(block, "norm1") else block.norm
        assert norm.linear.in_features == 256  # AdaLN input width = router input width
    for name in ("add_q_proj", "add_k_proj", "add_v_proj", "norm_added_q", "norm_added_k", "to_add_out", "to_out"):
        assert getattr(model.transformer_blocks[0].attn, name) is not None, name

    g = torch.Generator(device="cpu").manual_seed(1)
    T, te = 16, 11
    mask = torch.zeros((1, T), dtype=torch.bool)
    mask[:, :te] = True
    args = dict(hidden_states=torch.randn((1, 4, LATENT[0], 2 * LATENT[1], 2 * LATENT[2]), generator=g).to(dev(), torch.bfloat16),
                timestep=torch.tensor([500.0], device=dev()),
                encoder_hidden_states=torch.randn((1, T, 32), generator=g).to(dev(), torch.bfloat16),
                encoder_attention_mask=mask.to(dev()),
                pooled_projections=torch.randn((1, 16), generator=g).to(dev(), torch.bfloat16),
                guidance=torch.tensor([6000.0], device=dev(), dtype=torch.bfloat16), return_dict=False)
    with torch.no_grad():
        stock = model(**args)[0]
        kw = prepare_hunyuan_self_attn_kwargs(dict(latent_shape=LATENT, window_size=WINDOW, tile_size=TILE,
                                                   lowres_window_size=GROUP, lowres_reduction_rate=0.5), dev(), tau_sparse=0.3)
        apply_vorta_transformer(model, router_dtype=torch.bfloat16)
        assert all(hasattr(b, "router") for b in list(model.transformer_blocks) + list(model.single_transformer_blocks))
        _all_dense_routers(model)
        out = model(**args, self_attention_kwargs=kw, return_routing_scores=True)
        assert isinstance(out, tuple) and len(out) == 5 and len(out[4]) == 4 and out[4][0].shape == (1, 2, 3)
        routed_dense = out[0]
        _spread_routers(model, 3)
        routed = model(**args, self_attention_kwargs=kw)[0]
        assert routed.shape == stock.shape and torch.isfinite(routed.float()).all()
        apply_sp_flashattn_transformer(model)
        dense = model(**args)[0]
    _close(routed_dense, dense, 2e-3)  # the same kernel behind both entry points
    _close(dense, stock, 3e-2)         # HIP dense attention vs the stock processors (torch SDPA), 16-bit model


def test_wan_real_class_attribute_names_and_all_dense_equivalence():
    from diffusers import WanTransformer3DModel

    from vorta.patch.modeling_wan import apply_sp_flashattn_transformer, apply_vorta_transformer
    from vorta.patch.utils import prepare_wan_self_attn_kwargs
    torch.manual_seed(0)
    model = _build(WanTransformer3DModel, patch_size=(1, 2, 2), num_attention_heads=2, attention_head_dim=128, in_channels=4,
                   out_channels=4, text_dim=32, freq_dim=64, ffn_dim=512, num_layers=2, cross_attn_norm=True,
                   qk_norm="rms_norm_across_heads", eps=1e-6, rope_max_seq_len=64).to(dev()).to(torch.bfloat16).eval()
    assert len(model.blocks) == 2 and hasattr(model, "rope")
    assert hasattr(model.condition_embedder, "time_embedder") and model.condition_embedder.time_proj.in_features == 256
    for block in model.blocks:
        for attn in (block.attn1, block.attn2):
            assert attn.heads == 2 and callable(attn.set_processor)
            for name in ("to_q", "to_k", "to_v", "norm_q", "norm_k", "to_out"):
                assert getattr(attn, name) is not None, name

    g = torch.Generator(device="cpu").manual_seed(2)
    args = dict(hidden_states=torch.randn((1, 4, LATENT[0], 2 * LATENT[1], 2 * LATENT[2]), generator=g).to(dev(), torch.bfloat16),
                timestep=torch.tensor([500], device=dev()),
                encoder_hidden_states=torch.randn((1, 12, 32), generator=g).to(dev(), torch.bfloat16), return_dict=False)
    with torch.no_grad():
        stock = model(**args)[0]
        kw = prepare_wan_self_attn_kwargs(dict(latent_shape=LATENT, window_size=WINDOW, tile_size=TILE,
                                               lowres_window_size=GROUP, lowres_reduction_rate=0.5), dev(), tau_sparse=0.3)
        apply_vorta_transformer(model, router_dtype=torch.bfloat16)
        assert all(hasattr(b, "router") for b in model.blocks)
        _all_dense_routers(model)
        out = model(**args, self_attention_kwargs=kw, return_routing_scores=True)
        assert isinstance(out, tuple) and len(out) == 5 and len(out[4]) == 2 and out[4][0].shape == (1, 2, 3)
        routed_dense = out[0]
        _spread_routers(model, 4)
        routed = model(**args, self_attention_kwargs=kw)[0]
        assert routed.shape == stock.shape and torch.isfinite(routed.float()).all()
        apply_sp_flashattn_transformer(model)
        dense = model(**args)[0]
    _close(routed_dense, dense, 2e-3)
    _close(dense, stock, 3e-2)
