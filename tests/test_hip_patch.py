"""GPU: the model / pipeline entry points (vorta.patch.modeling_*, pipeline_*) on structural stand-ins of the
diffusers classes (tests/_mini_diffusers.py): the routes of every layer come from ONE route-plan call on the pure
timestep embedding, every block's attention gets its own layer's routes and the step's keyword set, the forward keeps
the reference's protocol, and each layer's attention output matches the oracle on the very q,k,v it was given."""
import os
import socket

import numpy as np
import pytest
import torch

from oracle import vorta_oracle as O
from _util import dev, rel_fro
import _mini_diffusers as M

pytestmark = pytest.mark.gpu

LATENT, TILE, WINDOW, GROUP = (4, 6, 8), (2, 3, 4), (3, 3, 3), (2, 3, 2)
S = 4 * 6 * 8
T, TE = 16, 11
TAU = 0.3


def _hy_kwargs():
    from vorta.patch.utils import prepare_hunyuan_self_attn_kwargs
    return prepare_hunyuan_self_attn_kwargs(dict(latent_shape=LATENT, window_size=WINDOW, tile_size=TILE,
                                                 lowres_window_size=GROUP, lowres_reduction_rate=0.5), dev(), TAU)


def _wan_kwargs():
    from vorta.patch.utils import prepare_wan_self_attn_kwargs
    return prepare_wan_self_attn_kwargs(dict(latent_shape=LATENT, window_size=WINDOW, tile_size=TILE,
                                             lowres_window_size=GROUP, lowres_reduction_rate=0.5), dev(), TAU)


def _spread_routers(model, seed):
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if type(m).__name__ == "Router":  # wide logits: no near-ties between experts in bf16
            m.linear.weight.data = (torch.randn(m.linear.weight.shape, generator=g) * 0.5).to(m.linear.weight)
            m.linear.bias.data = (torch.randn(m.linear.bias.shape, generator=g) * 2.0).to(m.linear.bias)


def _hy_model(seed=0, **kw):
    from vorta.patch.modeling_hunyuan import apply_vorta_transformer
    torch.manual_seed(seed)
    model = M.MiniHunyuanTransformer(**kw).to(dev()).to(torch.bfloat16)
    apply_vorta_transformer(model, router_dtype=torch.bfloat16)
    _spread_routers(model, seed + 1)
    return model


def _hy_inputs(seed=1):
    g = torch.Generator(device=dev()).manual_seed(seed)
    bf = torch.bfloat16
    mask = torch.zeros((1, T), device=dev())
    mask[:, :TE] = 1
    return dict(hidden_states=torch.randn((1, 4) + LATENT, device=dev(), generator=g).to(bf),
                timestep=torch.tensor([637.0], device=dev()),
                encoder_hidden_states=torch.randn((1, T, 24), device=dev(), generator=g).to(bf),
                encoder_attention_mask=mask,
                pooled_projections=torch.randn((1, 16), device=dev(), generator=g).to(bf),
                guidance=torch.tensor([6000.0], device=dev()).to(bf))


class _Recorder:
    """wraps `vorta_amd.routed.routed_attention` -- what torch.ops.vorta.routed_attention, the operator the processors launch
    through, runs (vorta_amd/torch_ops.py): records the inputs and the output view per call"""

    def __init__(self, module=None):
        import vorta_amd.routed as routed_module
        module = routed_module
        self.module, self.calls, self.stock = module, [], module.routed_attention

    def __enter__(self):
        def rec(q, k, v, routing, geom, **kw):
            out = self.stock(q, k, v, routing, geom, **kw)
            self.calls.append(dict(q=q.detach().clone(), k=k.detach().clone(), v=v.detach().clone(),
                                   lists=routing.lists.clone(), counts=routing.counts_dev.clone(),
                                   out=kw["out"].detach().clone(), kw=kw))
            return out
        self.module.routed_attention = rec
        return self

    def __exit__(self, *a):
        self.module.routed_attention = self.stock


def _experts_from(call, H):
    lists, counts = call["lists"].cpu().numpy(), call["counts"].cpu().numpy()
    experts = np.full(H, -1)
    for e in range(3):
        experts[lists[e, :counts[e]]] = e
    assert (experts >= 0).all()
    return experts


def _f64(t):
    return t.detach().double().cpu().numpy()


def test_hunyuan_transformer_routes_every_layer_from_one_plan():
    import vorta_amd.attention.hunyuan as hy
    model = _hy_model()
    blocks = list(model.transformer_blocks) + list(model.single_transformer_blocks)
    seen = []
    h = model.time_text_embed.timestep_embedder.register_forward_hook(lambda m, a, o: seen.append(o.detach().clone()))
    with _Recorder(hy) as rec:
        out = model(**_hy_inputs(), return_dict=False, self_attention_kwargs=_hy_kwargs(), return_routing_scores=True)
    h.remove()
    # the reference's routed forward protocol (modeling_hunyuan.py:441-442)
    assert isinstance(out, tuple) and len(out) == 5 and out[1] is None and out[2] is None and out[3] is None
    sample, scores = out[0], out[4]
    assert sample.shape == (1, 4) + LATENT and torch.isfinite(sample.float()).all()
    assert len(scores) == len(blocks) == len(rec.calls) == 4
    temb = seen[0]
    gi = O.group_info(LATENT, GROUP, 0.5)
    mixes = set()
    for layer, (block, call) in enumerate(zip(blocks, rec.calls)):
        # plan scores == the block's own Router module (vorta/patch/router.py:33-43), bit for bit
        assert torch.equal(scores[layer].to(dev()), block.router(temb))
        want = O.router_scores(_f64(temb), _f64(block.router.linear.weight), _f64(block.router.linear.bias), M.H)
        # bf16 module (silu, logits and scores rounded to bf16, router.py:41-43) vs float64: stated tolerance 2e-2
        assert np.abs(scores[layer].double().numpy() - want).max() < 2e-2
        experts = O.route_heads(scores[layer].double().numpy(), TAU)  # the top-1 / tau rule on the module's scores
        assert (_experts_from(call, M.H) == experts).all(), (layer, experts)
        mixes.add(tuple(experts))
        assert call["kw"]["text_len"] == T and call["kw"]["text_valid"] == TE
        ref = O.routed_attention(_f64(call["q"]), _f64(call["k"]), _f64(call["v"]), experts, model="hunyuan",
                                 latent=LATENT, tile=TILE, window=WINDOW, gi=gi, t_text=T, t_eff=TE)
        got = call["out"].float().cpu().numpy()
        assert rel_fro(got, ref) < 1e-2, layer
        assert (got[:, :, S + TE:] == 0).all()  # padded text rows (hunyuan.py:176)
    assert len(mixes) > 1, "routers of different layers should disagree in this fixture"
    # dict output + scores off by default
    out2 = model(**_hy_inputs(), self_attention_kwargs=_hy_kwargs())
    assert torch.equal(out2.sample, sample) and out2.routing_scores == [] and out2["sample"] is out2.sample


def test_hunyuan_plan_path_equals_per_layer_dispatch():
    """Feeding each processor only `routing_score` (the reference's per-block route, hunyuan.py:612-640) gives the
    same bits as the plan's device-resident head lists."""
    from vorta_amd.patch import _engine as E
    model = _hy_model(seed=3)
    a = model(**_hy_inputs(), return_dict=False, self_attention_kwargs=_hy_kwargs())[0]
    blocks = list(model.transformer_blocks) + list(model.single_transformer_blocks)
    for b in blocks:
        b.attn.processor._accepted = b.attn.processor._accepted - {"head_routing", "experts_host"}
    b_out = model(**_hy_inputs(), return_dict=False, self_attention_kwargs=_hy_kwargs())[0]
    assert torch.equal(a, b_out)
    assert isinstance(blocks[0].attn.processor, E.BoundProcessor)


def test_hunyuan_token_replace_uses_first_timestep_embedding_and_descriptor_is_cached():
    from vorta_amd.patch import _engine as E
    model = _hy_model(seed=5, token_replace=True)
    calls = []
    h = model.time_text_embed.timestep_embedder.register_forward_hook(lambda m, a, o: calls.append(o.detach().clone()))
    inp = _hy_inputs()
    out = model(**inp, return_dict=False, self_attention_kwargs=_hy_kwargs(), return_routing_scores=True)
    assert len(calls) == 2  # real timestep, then the zero timestep of token_replace (modeling_hunyuan.py:633-637)
    block = model.transformer_blocks[0]
    assert torch.equal(out[4][0].to(dev()), block.router(calls[0]))
    assert not torch.equal(block.router(calls[0]), block.router(calls[1]))
    ctx = E.context_of(model)
    desc = ctx.descriptor_cache["entry"][3]
    assert desc.text_seq_length == T and desc.text_seq_length_no_pad == TE
    model(**inp, return_dict=False, self_attention_kwargs=_hy_kwargs())
    assert ctx.descriptor_cache["entry"][3] is desc  # same prompt tensor: no rebuild, no host read
    h.remove()
    with pytest.raises(NotImplementedError):
        model(**inp, self_attention_kwargs=_hy_kwargs(), return_losses=True)
    with pytest.raises(ValueError):
        model(**inp)


def test_hunyuan_descriptor_follows_the_prompt_not_the_address():
    """two prompts with different valid text lengths through one patched model; the first prompt's mask is freed before
    the second is created, so the allocator may hand out the same address (a cache keyed on data_ptr served
    prompt 1's valid length to prompt 2)"""
    from vorta_amd.patch import _engine as E
    from vorta.attention import create_sliding_tile_attn_mask_func
    model = _hy_model(seed=9)
    ctx = E.context_of(model)
    inp = _hy_inputs()
    model(**inp, return_dict=False, self_attention_kwargs=_hy_kwargs())
    assert ctx.descriptor_cache["entry"][3].text_seq_length_no_pad == TE
    ptr = inp["encoder_attention_mask"].data_ptr()
    del inp["encoder_attention_mask"]
    # worst case: nothing holds the old mask any more (the entry normally does), so its address is free for reuse
    ctx.descriptor_cache["entry"] = (None,) + ctx.descriptor_cache["entry"][1:]
    te2 = 5
    mask2 = torch.zeros((1, T), device=dev())
    mask2[:, :te2] = 1
    inp["encoder_attention_mask"] = mask2
    out = model(**inp, return_dict=False, self_attention_kwargs=_hy_kwargs())[0]
    assert ctx.descriptor_cache["entry"][0] is mask2 and ctx.descriptor_cache["entry"][3].text_seq_length_no_pad == te2
    want = create_sliding_tile_attn_mask_func(latent_shape=LATENT, window_size=WINDOW, tile_size=TILE, text_seq_length=T,
                                              text_seq_length_no_pad=te2, device=dev())
    ref = model(**inp, return_dict=False, self_attention_kwargs=dict(_hy_kwargs(), flex_attn_mask_func=want))[0]
    assert torch.equal(out, ref)
    print("mask address reused by the allocator:", mask2.data_ptr() == ptr)
    # in-place edits of the same tensor are seen through its version counter
    mask2[:, :8] = 1
    model(**inp, return_dict=False, self_attention_kwargs=_hy_kwargs())
    assert ctx.descriptor_cache["entry"][3].text_seq_length_no_pad == 8


def test_wan_rope_table_follows_the_frequency_tensor_not_its_address():
    """480x832 then 832x480 in one process: equal shapes, different contents, and the freed table's address is reused"""
    from vorta_amd.attention.wan import _cos_sin
    rope = M.MiniWanRope()
    def table(f, h, w):
        return rope(torch.zeros((1, 4, f, h, w), device=dev()))
    f1 = table(4, 6, 8)
    c1, s1 = _cos_sin(f1)
    assert _cos_sin(f1)[0] is c1  # the blocks of one forward share the entry
    ptr = f1.data_ptr()
    del f1
    import vorta_amd.attention.wan as wan
    wan._ROPE_CACHE["entry"] = (None,) + wan._ROPE_CACHE["entry"][1:]
    f2 = table(4, 8, 6)
    c2, s2 = _cos_sin(f2)
    flat = f2.reshape(-1, f2.shape[-1])
    assert torch.equal(c2, flat.real.float().repeat_interleave(2, dim=1))
    assert torch.equal(s2, flat.imag.float().repeat_interleave(2, dim=1))
    assert not torch.equal(c2, c1)
    print("frequency table address reused by the allocator:", f2.data_ptr() == ptr)


def test_hunyuan_dense_mask_shapes():
    """L = valid keys from a [B,1,1,N] key mask (diffusers 0.33) or one row of a [B,1,N,N] / [B,N,N] mask; other
    shapes are refused"""
    from vorta_amd.attention.hunyuan import _valid_keys
    N, L = 40, 29
    key = torch.zeros((1, 1, 1, N), dtype=torch.bool, device=dev())
    key[..., :L] = True
    assert int(_valid_keys(key)) == L
    assert int(_valid_keys(key.expand(1, 1, N, N))) == L
    assert int(_valid_keys(key[0].expand(1, N, N))) == L
    assert int(_valid_keys(key.float())) == L
    with pytest.raises(ValueError):
        _valid_keys(torch.ones((1, 1, 3, N), device=dev()))


def test_hunyuan_native_attention_patch_matches_all_dense_routing():
    from vorta.patch.modeling_hunyuan import apply_sp_flashattn_transformer
    model = _hy_model(seed=7)
    kw = dict(_hy_kwargs(), tau_sparse=1.1)  # every top-1 score is below tau: expert 0 for all heads
    a = model(**_hy_inputs(), return_dict=False, self_attention_kwargs=kw)[0]
    apply_sp_flashattn_transformer(model)
    out = model(**_hy_inputs(), return_dict=False)
    assert isinstance(out, tuple) and len(out) == 1  # stock protocol again
    assert torch.equal(out[0], a)


def _wan_model(seed=0):
    from vorta.patch.modeling_wan import apply_vorta_transformer
    torch.manual_seed(seed)
    model = M.MiniWanTransformer().to(dev()).to(torch.bfloat16)
    apply_vorta_transformer(model, router_dtype=torch.bfloat16)
    _spread_routers(model, seed + 1)
    return model


def _wan_inputs(seed=2):
    g = torch.Generator(device=dev()).manual_seed(seed)
    return dict(hidden_states=torch.randn((1, 4) + LATENT, device=dev(), generator=g).to(torch.bfloat16),
                timestep=torch.tensor([412.0], device=dev()),
                encoder_hidden_states=torch.randn((1, 20, 24), device=dev(), generator=g).to(torch.bfloat16))


def test_wan_transformer_routes_every_layer_from_one_plan():
    import vorta_amd.attention.wan as wan
    model = _wan_model()
    seen = []
    h = model.condition_embedder.time_embedder.register_forward_hook(lambda m, a, o: seen.append(o.detach().clone()))
    with _Recorder(wan) as rec:
        out = model(**_wan_inputs(), return_dict=False, self_attention_kwargs=_wan_kwargs(), return_routing_scores=True)
    h.remove()
    assert len(out) == 5 and len(out[4]) == len(model.blocks) == len(rec.calls) == 3
    gi = O.group_info(LATENT, GROUP, 0.5)
    for layer, (block, call) in enumerate(zip(model.blocks, rec.calls)):
        assert torch.equal(out[4][layer].to(dev()), block.router(seen[0]))
        want = O.router_scores(_f64(seen[0]), _f64(block.router.linear.weight), _f64(block.router.linear.bias), M.H)
        assert np.abs(out[4][layer].double().numpy() - want).max() < 2e-2
        experts = O.route_heads(out[4][layer].double().numpy(), TAU)
        assert (_experts_from(call, M.H) == experts).all()
        ref = O.routed_attention(_f64(call["q"]), _f64(call["k"]), _f64(call["v"]), experts, model="wan",
                                 latent=LATENT, tile=TILE, window=WINDOW, gi=gi)
        assert rel_fro(call["out"].float().cpu().numpy(), ref) < 1e-2, layer
    # cross attention went through the dense sequence-parallel-aware processor (modeling_wan.py:299)
    from vorta_amd.attention import WanAttnProcessor2_0
    assert type(model.blocks[0].attn2.processor) is WanAttnProcessor2_0


def test_route_plan_is_graph_capturable_and_replays_new_routes():
    """the plan + every layer's routed attention enqueue without a host sync, so one hipGraph
    holds a whole step's attention schedule and replays against the routes of a new timestep."""
    from vorta_amd.patch._engine import RoutePlan
    from vorta_amd.patch.router import Router
    from vorta_amd.routed import geometry_for, routed_attention
    torch.manual_seed(11)
    L, E_dim, H = 3, 256, 6
    routers = [Router(E_dim, H).to(dev()).to(torch.bfloat16) for _ in range(L)]
    for r in routers:
        r.linear.bias.data.normal_(0, 2.0)
    plan = RoutePlan(routers)
    geom = geometry_for((8, 6, 8), (2, 3, 4), WINDOW, GROUP, 0.5, dev())
    Sg = 8 * 6 * 8
    q, k, v = (torch.randn((1, H, Sg + T, 128), device=dev()).to(torch.bfloat16) for _ in range(3))
    temb = torch.randn((1, E_dim), device=dev()).to(torch.bfloat16)
    outs = [torch.empty_like(q) for _ in range(L)]

    def step():
        plan.compute(temb, TAU)
        for layer in range(L):
            routed_attention(q, k, v, plan.routing(layer), geom, model="hunyuan", text_len=T, text_valid=TE,
                             out=outs[layer])

    step()  # warm-up: tables, buffers
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    first = plan._out[1].clone()
    temb.copy_(torch.randn((1, E_dim), device=dev()).to(torch.bfloat16) * 3)
    graph.replay()
    torch.cuda.synchronize()
    replayed = [o.clone() for o in outs]
    routes = plan._out[1].clone()
    assert not torch.equal(first, routes), "new timestep embedding should change some routes"
    step()  # eager, same inputs
    torch.cuda.synchronize()
    assert torch.equal(routes, plan._out[1])
    for a, b in zip(replayed, outs):
        assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------------- pipelines
def _hy_pipe_inputs():
    inp = _hy_inputs()
    return dict(prompt_embeds=inp["encoder_hidden_states"], pooled_prompt_embeds=inp["pooled_projections"],
                prompt_attention_mask=inp["encoder_attention_mask"], height=LATENT[1], width=LATENT[2],
                num_frames=LATENT[0], num_inference_steps=3)


def test_hunyuan_pipeline_call():
    from vorta.patch import _pipeline as P
    from vorta.patch.pipeline_hunyuan import sp_pipeline_call, vorta_pipeline_call
    from vorta.patch.modeling_hunyuan import apply_sp_flashattn_transformer

    class Pipe(M.MiniHunyuanPipeline):
        pass

    P.register_pipeline_class(Pipe)
    stock = Pipe.__call__
    Pipe.__call__ = vorta_pipeline_call  # scripts/hunyuan/inference.py:120
    model = _hy_model(seed=9)
    pipe = Pipe(model, dev())
    gen = torch.Generator(device=dev())
    video, scores = pipe(**_hy_pipe_inputs(), generator=gen.manual_seed(4), output_type="latent", return_dict=False,
                         self_attention_kwargs=_hy_kwargs(), return_routing_scores=True)
    assert video.shape == (1, 4) + LATENT
    assert len(scores) == 3 and all(len(s) == 4 and s[0].shape == (1, M.H, 3) for s in scores)
    assert not torch.equal(scores[0][0], scores[2][0])  # routes follow the timestep
    # the same loop written out with explicit keywords on the transformer gives the same latents
    Pipe.__call__ = stock
    kw = _hy_kwargs()

    class Explicit(torch.nn.Module):
        dtype, config = model.dtype, model.config

        def forward(self, **k):
            return model(**k, self_attention_kwargs=kw)

    ref = Pipe(Explicit(), dev())(**_hy_pipe_inputs(), generator=gen.manual_seed(4), output_type="latent",
                                  return_dict=False)[0]
    assert torch.equal(video, ref)
    # record form, scores off, decoded output
    Pipe.__call__ = vorta_pipeline_call
    out = pipe(**_hy_pipe_inputs(), generator=gen.manual_seed(4), self_attention_kwargs=_hy_kwargs())
    assert out.routing_scores is None and isinstance(out.frames, np.ndarray)
    assert np.allclose(out.frames, (video / 0.5 * 2.0).float().cpu().numpy(), atol=1e-5)
    # native attention: sp_pipeline_call accepts and ignores self_attention_kwargs (pipeline_hunyuan.py:63)
    Pipe.__call__ = sp_pipeline_call
    apply_sp_flashattn_transformer(model)
    v2, none = pipe(**_hy_pipe_inputs(), generator=gen.manual_seed(4), output_type="latent", return_dict=False,
                    self_attention_kwargs=None)
    assert none is None and v2.shape == video.shape


def test_wan_pipeline_call_records_conditional_forward_only():
    from vorta.patch import _pipeline as P
    from vorta.patch import _engine as E
    from vorta.patch.pipeline_wan import vorta_pipeline_call

    class Pipe(M.MiniWanPipeline):
        pass

    P.register_pipeline_class(Pipe)
    Pipe.__call__ = vorta_pipeline_call
    model = _wan_model(seed=13)
    pipe = Pipe(model, dev())
    inp = _wan_inputs()
    neg = torch.randn_like(inp["encoder_hidden_states"])
    before = E.context_of(model).forwards
    video, scores = pipe(prompt_embeds=inp["encoder_hidden_states"], negative_prompt_embeds=neg, height=LATENT[1],
                         width=LATENT[2], num_frames=LATENT[0], num_inference_steps=2, output_type="latent",
                         generator=torch.Generator(device=dev()).manual_seed(1), return_dict=False,
                         self_attention_kwargs=_wan_kwargs(), return_routing_scores=True)
    assert E.context_of(model).forwards - before == 4  # two batch-1 forwards per step (pipeline_wan.py:322-344)
    assert len(scores) == 2 and all(len(s) == 3 for s in scores)
    assert torch.isfinite(video).all()


# ------------------------------------------------------------------------------ sequence parallel, rehearsal
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _sp_pipe_worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vorta.patch import _pipeline as P
    from vorta.patch.pipeline_hunyuan import vorta_pipeline_call
    from vorta_amd.ulysses import SP_STATE

    class Pipe(M.MiniHunyuanPipeline):
        pass

    P.register_pipeline_class(Pipe)
    Pipe.__call__ = vorta_pipeline_call
    model = _hy_model(seed=9)
    pipe = Pipe(model, dev())
    gen = torch.Generator(device=dev())
    args = dict(_hy_pipe_inputs(), output_type="np", return_dict=False, self_attention_kwargs=_hy_kwargs(),
                return_routing_scores=True)
    full, s_full = pipe(**args, generator=gen.manual_seed(4))
    SP_STATE.setup_sp_group(world)
    part, s_part = pipe(**args, generator=gen.manual_seed(4))
    # every rank ends with the whole video (the reference all-gathers frame shards, pipeline_hunyuan.py:453-454)
    ret[rank] = (float(np.abs(part - full).max()), float(np.abs(full).max()),
                 all(torch.equal(a, b) for x, y in zip(s_full, s_part) for a, b in zip(x, y)))
    # no generator: the ranks agree on a seed (pipeline_hunyuan.py:76-83)
    a = pipe(**dict(args, output_type="latent"))[0]
    gathered = [torch.empty_like(a.cpu()) for _ in range(world)]
    dist.all_gather(gathered, a.cpu())
    ret[rank] += (all(torch.equal(gathered[0], g) for g in gathered),)
    dist.barrier()
    SP_STATE.cleanup()


def _sp_pipe_odd_frames_worker(rank, world, port, ret):
    """3 latent frames on 2 ranks: the reference's frame shard refuses this (pipeline_hunyuan.py:367-369 needs
    frames % P == 0); the token-level shard cuts the 144 tokens into 2 x 72"""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vorta.patch import _pipeline as P
    from vorta.patch.pipeline_hunyuan import vorta_pipeline_call
    from vorta.patch.utils import prepare_hunyuan_self_attn_kwargs
    from vorta_amd.ulysses import SP_STATE

    class Pipe(M.MiniHunyuanPipeline):
        pass

    P.register_pipeline_class(Pipe)
    Pipe.__call__ = vorta_pipeline_call
    lat = (3, 6, 8)
    kw = prepare_hunyuan_self_attn_kwargs(dict(latent_shape=lat, window_size=WINDOW, tile_size=(1, 3, 4),
                                               lowres_window_size=(3, 3, 2), lowres_reduction_rate=0.5), dev(), TAU)
    model = _hy_model(seed=9)
    pipe = Pipe(model, dev())
    gen = torch.Generator(device=dev())
    inp = _hy_pipe_inputs()
    inp.update(height=lat[1], width=lat[2], num_frames=lat[0])
    args = dict(inp, output_type="latent", return_dict=False, self_attention_kwargs=kw, return_routing_scores=True)
    full, s_full = pipe(**args, generator=gen.manual_seed(4))
    SP_STATE.setup_sp_group(world)
    part, s_part = pipe(**args, generator=gen.manual_seed(4))
    ret[rank] = (float((part - full).abs().max()), float(full.abs().max()), tuple(part.shape) == (1, 4) + lat,
                 all(torch.equal(a, b) for x, y in zip(s_full, s_part) for a, b in zip(x, y)))
    dist.barrier()
    SP_STATE.cleanup()


@pytest.mark.parametrize("world", [2, 4])
def test_hunyuan_pipeline_token_shard_takes_any_frame_count(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_sp_pipe_odd_frames_worker, args=(r, world, port, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=600)
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        for r in range(world):
            err, mag, shape_ok, same_scores = ret[r]
            assert err <= 2e-2 * max(mag, 1.0) and shape_ok and same_scores, (r, ret[r])


@pytest.mark.parametrize("precision", ["native", "auto8"])
def test_hunyuan_pipeline_under_sequence_parallel_rehearsal(precision, monkeypatch):
    """2 ranks sharing the one GPU (gloo, host-staged transport): whole latents on every rank, token-sharded inside the
    transformer, global rotary table and attention mask from the stock forward, video == single process.  With
    VORTA_ATTENTION_PRECISION=auto8 the unchanged call goes through the 8-bit kernels (text rows and padding through the
    segmented quantiser, v as e4m3 on the wire): the ranks still agree bit for bit; against the single-process 8-bit run
    the video differs by another realisation of the rounding noise (the key centres are means of other sample rows)."""
    import torch.multiprocessing as mp
    if precision != "native":
        monkeypatch.setenv("VORTA_ATTENTION_PRECISION", precision)
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_sp_pipe_worker, args=(r, 2, port, ret)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=600)
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        for r in range(2):
            err, mag, same_scores, same_seed = ret[r]
            assert err <= (2e-2 if precision == "native" else 0.25) * max(mag, 1.0), (r, err, mag)
            assert same_scores and same_seed


def test_train_router_patch_runs_the_soft_mixture_forward():
    """apply_vorta_transformer(train_router=True): Train processors, no tau in the keyword set (the reference's training
    config has none), every layer = score-weighted sum of the three experts; gradients are refused, not dropped."""
    import vorta_amd.attention.hunyuan as hy
    from vorta.patch.modeling_hunyuan import apply_vorta_transformer
    from vorta.patch.utils import prepare_hunyuan_self_attn_kwargs
    torch.manual_seed(21)
    model = M.MiniHunyuanTransformer().to(dev()).to(torch.bfloat16)
    apply_vorta_transformer(model, train_router=True, router_dtype=torch.bfloat16)
    _spread_routers(model, 22)
    kw = prepare_hunyuan_self_attn_kwargs(dict(latent_shape=LATENT, window_size=WINDOW, tile_size=TILE,
                                               lowres_window_size=GROUP, lowres_reduction_rate=0.5), dev())
    assert "tau_sparse" not in kw
    calls = []
    import vorta_amd.routed as rt  # the processors go through torch.ops.vorta.soft_mixture_attention, which lands here
    stock = rt.soft_mixture_attention

    def rec(q, k, v, scores, geom, **k2):
        out = stock(q, k, v, scores, geom, **k2)
        calls.append((q.detach().clone(), k.detach().clone(), v.detach().clone(), scores.detach().clone(), out.detach().clone(), k2))
        return out

    rt.soft_mixture_attention = rec
    try:
        with torch.no_grad():
            out = model(**_hy_inputs(), return_dict=False, self_attention_kwargs=kw, return_routing_scores=True)
    finally:
        rt.soft_mixture_attention = stock
    assert len(out) == 5 and len(calls) == 4 and torch.isfinite(out[0].float()).all()
    gi = O.group_info(LATENT, GROUP, 0.5)
    for layer, (q, k, v, sc, o, k2) in enumerate(calls):
        assert torch.equal(sc.cpu(), out[4][layer])
        ref = O.soft_mixture_attention(_f64(q), _f64(k), _f64(v), _f64(sc), model="hunyuan", latent=LATENT, tile=TILE,
                                       window=WINDOW, gi=gi, t_text=T, t_eff=TE)
        assert rel_fro(o.float().cpu().numpy(), ref) < 1e-2, layer
    with pytest.raises(NotImplementedError):  # grad mode on + trainable router scores would need a backward
        inp = _hy_inputs()
        inp["hidden_states"].requires_grad_(True)
        model(**inp, return_dict=False, self_attention_kwargs=kw)
