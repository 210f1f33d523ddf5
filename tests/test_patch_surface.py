"""CPU: the model / pipeline entry points of vorta.patch (SURVEY.md §8b B-py) up to the GPU boundary -- what
gets mounted where, the checkpoint key format, the forward / call protocols, the sequence-parallel latent shard --
on structural stand-ins of the diffusers classes (tests/_mini_diffusers.py)."""
import os
import socket

import numpy as np
import pytest
import torch

import _mini_diffusers as M

KW = dict(latent_shape=(4, 6, 8), window_size=(3, 3, 3), tile_size=(2, 3, 4), lowres_window_size=(2, 3, 2),
          lowres_reduction_rate=0.5)


def _inputs():
    torch.manual_seed(0)
    return dict(hidden_states=torch.randn(1, 4, 4, 6, 8), timestep=torch.tensor([500.0]),
                encoder_hidden_states=torch.randn(1, 16, 24), encoder_attention_mask=torch.ones(1, 16),
                pooled_projections=torch.randn(1, 16))


def test_reference_import_sites_resolve():
    """scripts/hunyuan/inference.py:27-32, scripts/wan/inference.py:32-37"""
    from vorta.patch.modeling_hunyuan import apply_sp_flashattn_transformer, apply_vorta_transformer  # noqa: F401
    from vorta.patch.pipeline_hunyuan import sp_pipeline_call, vorta_pipeline_call  # noqa: F401
    from vorta.patch.utils import hunyuan_pixel2token, prepare_hunyuan_self_attn_kwargs  # noqa: F401
    import vorta.patch.modeling_wan as mw
    import vorta.patch.pipeline_wan as pw
    from vorta.patch.utils import prepare_wan_self_attn_kwargs, wan_pixel2token  # noqa: F401
    for mod, names in ((mw, ("apply_sp_flashattn_transformer", "apply_vorta_transformer")),
                       (pw, ("sp_pipeline_call", "vorta_pipeline_call", "apply_vorta_pipeline"))):
        for n in names:
            assert callable(getattr(mod, n))
    from vorta.patch.outputs import RoutedTransformerModelOutput, VideoPipelineOutput
    r = RoutedTransformerModelOutput(sample=torch.zeros(1), routing_scores=[])
    assert r.to_tuple()[0] is r.sample and r["routing_scores"] == [] and r.reg_loss is None
    assert VideoPipelineOutput(frames=1).routing_scores is None


def test_apply_vorta_transformer_mounts_routers_processors_and_hooks(tmp_path):
    from vorta.patch import _engine as E
    from vorta.patch.modeling_hunyuan import apply_sp_flashattn_transformer, apply_vorta_transformer
    from vorta.attention import HunyuanVideoFlashAttnProcessor, HunyuanVideoFlashAttnProcessorTripleEval
    model = M.MiniHunyuanTransformer().to(torch.bfloat16)
    assert apply_vorta_transformer(model, router_dtype=torch.bfloat16) is model
    blocks = list(model.transformer_blocks) + list(model.single_transformer_blocks)
    for i, b in enumerate(blocks):
        assert b.router.linear.in_features == M.INNER and b.router.linear.out_features == 3 * M.H
        assert b.router.linear.weight.dtype == torch.bfloat16
        p = b.attn.processor
        assert isinstance(p, E.BoundProcessor) and p.layer == i
        assert type(p.inner) is HunyuanVideoFlashAttnProcessorTripleEval
        assert p.inner.check_input == (i == 0)  # modeling_hunyuan.py:678,684
    # the published checkpoints' key format (vorta/train/checkpoint.py:63-73: every key containing 'router')
    keys = [k for k in model.state_dict() if "router" in k]
    assert "transformer_blocks.0.router.linear.weight" in keys and "single_transformer_blocks.1.router.linear.bias" in keys
    assert len(keys) == 2 * len(blocks)
    n_hooks = len(model._vorta_hooks)
    apply_vorta_transformer(model)  # idempotent: routers kept, hooks replaced
    assert len(model._vorta_hooks) == n_hooks and len(E.context_of(model).plan) == len(blocks)
    # router-only checkpoint round trip
    ckpt = tmp_path / "router.pt"
    torch.save({k: torch.full_like(v, 0.25) for k, v in model.state_dict().items() if "router" in k}, ckpt)
    other = M.MiniHunyuanTransformer().to(torch.bfloat16)
    apply_vorta_transformer(other, checkpoint_file=ckpt, router_dtype=torch.bfloat16)
    assert all(torch.all(b.router.linear.weight == 0.25) for b in other.transformer_blocks)
    with pytest.raises(FileNotFoundError):
        apply_vorta_transformer(M.MiniHunyuanTransformer(), checkpoint_file=tmp_path / "missing.pt")
    # native attention patch
    apply_sp_flashattn_transformer(model)
    assert all(type(b.attn.processor) is HunyuanVideoFlashAttnProcessor for b in blocks)


def test_forward_protocol_fails_loudly_without_a_gpu():
    from vorta.patch.modeling_hunyuan import apply_vorta_transformer
    from vorta.patch.utils import prepare_hunyuan_self_attn_kwargs
    from vorta_amd._C import VortaHipError
    model = M.MiniHunyuanTransformer().to(torch.bfloat16)
    apply_vorta_transformer(model, router_dtype=torch.bfloat16)
    inp = {k: (v.to(torch.bfloat16) if v.is_floating_point() and k != "timestep" else v) for k, v in _inputs().items()}
    with pytest.raises(ValueError, match="self_attention_kwargs"):
        model(**inp)
    with pytest.raises(NotImplementedError):
        model(**inp, self_attention_kwargs={}, return_losses=True)
    kw = prepare_hunyuan_self_attn_kwargs(dict(KW), torch.device("cpu"), 0.3)
    assert "lowres_window_size" not in kw and kw["tau_sparse"] == 0.3
    with pytest.raises(VortaHipError):  # no CPU path behind the processors
        model(**inp, self_attention_kwargs=kw)
    with pytest.raises(ValueError, match="tau_sparse"):
        model(**inp, self_attention_kwargs={k: v for k, v in kw.items() if k != "tau_sparse"})


def test_wan_patch_mounts_on_attn1_and_dense_on_attn2():
    from vorta.patch import _engine as E
    from vorta.patch.modeling_wan import apply_vorta_transformer
    from vorta.attention import WanAttnProcessor2_0, WanAttnProcessorTripleEval
    model = M.MiniWanTransformer()
    apply_vorta_transformer(model, router_dtype=torch.bfloat16)
    for i, b in enumerate(model.blocks):
        assert b.router.linear.in_features == model.condition_embedder.time_proj.in_features
        assert isinstance(b.attn1.processor, E.BoundProcessor) and type(b.attn1.processor.inner) is WanAttnProcessorTripleEval
        assert type(b.attn2.processor) is WanAttnProcessor2_0
    assert "blocks.0.router.linear.weight" in model.state_dict()


class _PointwiseTransformer(torch.nn.Module):
    """per-token stand-in (no attention): lets the pipeline mechanics run on the CPU"""

    def __init__(self):
        super().__init__()
        self.config = type("C", (), dict(in_channels=4))()
        self.w = torch.nn.Parameter(torch.tensor(0.5))

    @property
    def dtype(self):
        return self.w.dtype

    def forward(self, hidden_states, timestep, **kw):
        return (torch.tanh(hidden_states * self.w) * (timestep.view(-1, 1, 1, 1, 1) / 1000.0),)


def _pipe_args():
    return dict(prompt_embeds=torch.zeros(1, 16, 24), pooled_prompt_embeds=torch.zeros(1, 16),
                prompt_attention_mask=torch.ones(1, 16), height=6, width=8, num_frames=4, num_inference_steps=3)


def test_pipeline_call_protocol_on_cpu():
    from vorta.patch import _pipeline as P
    from vorta.patch.pipeline_hunyuan import sp_pipeline_call, vorta_pipeline_call
    from vorta.patch.outputs import VideoPipelineOutput

    class Pipe(M.MiniHunyuanPipeline):
        pass

    class Unregistered:
        transformer = _PointwiseTransformer()
        __call__ = vorta_pipeline_call

    with pytest.raises(RuntimeError, match="register_pipeline_class"):
        Unregistered()(num_frames=4)
    P.register_pipeline_class(Pipe)
    stock = Pipe.__call__
    pipe = Pipe(_PointwiseTransformer(), "cpu")
    g = torch.Generator()
    want = pipe(**_pipe_args(), generator=g.manual_seed(3), return_dict=False)[0]
    Pipe.__call__ = vorta_pipeline_call
    P.register_pipeline_class(Pipe)  # registering again after the swap keeps the stock call
    assert P.original_call(pipe) is stock
    video, scores = pipe(**_pipe_args(), generator=g.manual_seed(3), return_dict=False, self_attention_kwargs=None)
    assert scores is None and np.array_equal(video, want)
    out = pipe(**_pipe_args(), generator=g.manual_seed(3))
    assert isinstance(out, VideoPipelineOutput) and np.array_equal(out.frames, want)
    with pytest.raises(TypeError):
        pipe(**_pipe_args(), no_such_keyword=1)
    Pipe.__call__ = sp_pipeline_call
    assert np.array_equal(pipe(**_pipe_args(), generator=g.manual_seed(3), return_dict=False,
                               self_attention_kwargs={"ignored": 1})[0], want)
    assert "prepare_latents" not in vars(pipe)


# ---------------------------------------------------------------------------- sequence parallel (gloo, 2 ranks)
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _sp_worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vorta.patch import _engine as E
    from vorta.patch import _pipeline as P
    from vorta.patch.pipeline_hunyuan import vorta_pipeline_call
    from vorta.ulysses import SP_STATE

    class Pipe(M.MiniHunyuanPipeline):
        pass

    P.register_pipeline_class(Pipe)
    pipe = Pipe(_PointwiseTransformer(), "cpu")
    g = torch.Generator()
    want = pipe(**_pipe_args(), generator=g.manual_seed(3), return_dict=False)[0]
    Pipe.__call__ = vorta_pipeline_call
    SP_STATE.setup_sp_group(world)
    shapes = []
    h = pipe.transformer.register_forward_pre_hook(lambda m, a, k: shapes.append(tuple(k["hidden_states"].shape)),
                                                   with_kwargs=True)
    got = pipe(**_pipe_args(), generator=g.manual_seed(3), return_dict=False)[0]
    h.remove()
    lat = pipe(**_pipe_args(), output_type="latent", return_dict=False)[0]  # no generator: seed agreed over the group
    gathered = [torch.empty_like(lat) for _ in range(world)]
    dist.all_gather(gathered, lat)
    # the stock rope sees the GLOBAL frame count through the zero-stride stand-in, and reads nothing from it
    rope = M.MiniHunyuanRope()
    model = torch.nn.Module()
    model._vorta_hooks = []
    E.install_sp_rope(rope, model)
    cos, _ = rope(torch.zeros(1, 4, 2, 6, 8))
    ret[rank] = dict(equal=bool(np.array_equal(got, want)), shapes=shapes, same_seed=all(torch.equal(gathered[0], x) for x in gathered),
                     rope_rows=cos.shape[0], freed=pipe.freed)
    dist.barrier()
    SP_STATE.cleanup()


def test_pipeline_call_keeps_whole_latents_under_sp():
    """pipeline calls under sequence parallelism: every rank carries the WHOLE latent (the transformer shards its token
    sequence itself, tests/test_ulysses_gloo.py::test_pipeline_token_shard_33_frames); a transformer called directly with
    frame-sharded latents still gets the global rotary table through the zero-stride stand-in"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_sp_worker, args=(r, 2, port, ret)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=300)
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        for r in range(2):
            d = ret[r]
            assert d["equal"], "the video must equal the single-process one"
            assert d["shapes"] == [(1, 4, 4, 6, 8)] * 3  # every forward of the loop sees all 4 latent frames
            assert d["same_seed"] and d["rope_rows"] == 4 * 6 * 8
