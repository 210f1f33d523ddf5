"""Shared machinery of the model / pipeline patches (vorta/patch/modeling_*.py, pipeline_*.py).

The reference threads three things through re-stated copies of the diffusers forwards: the pure timestep
embedding into every block's Router (modeling_hunyuan.py:206,350,491,555; modeling_wan.py:77,127,215), the
`self_attention_kwargs` into every self-attention call (modeling_hunyuan.py:492-499; modeling_wan.py:131-137), and the
routing scores back out (modeling_hunyuan.py:296-297,441-449).  This build keeps the stock diffusers forwards and
attaches to them with module hooks instead:

  * a forward hook on the timestep embedder captures the embedding once per forward and computes the routes of
    ALL layers in one call (vorta_route_plan: 2 launches per step instead of ~5 per layer);
  * every block's attention processor is a `BoundProcessor` that adds its layer's routes + the step's keyword
    set to the stock call `proc(attn, hidden_states, encoder_hidden_states, attention_mask, rotary)`;
  * a pre-hook / hook pair on the transformer accepts and strips the extra keywords of the reference's routed
    forward (`self_attention_kwargs`, `return_routing_scores`, `return_losses`, ...) and shapes the output like
    modeling_hunyuan.py:441-449.

Nothing is patched at class level, so two differently patched models can live in one process (the reference
overwrites `Class.forward`, SURVEY.md §8b "not re-entrant").
"""
from __future__ import annotations

import inspect
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence

import torch
from torch import nn

from .. import ops
from ..routed import HeadRouting
from ..ulysses import SP_STATE

_EXTRA_FORWARD_KEYS = ("self_attention_kwargs", "return_losses", "reture_hidden_layer_distill_loss",
                       "return_routing_scores")


class RoutePlan:
    """Routes of every layer for the current denoising step, resident on the device.

    The router input is the same timestep embedding for all blocks, so one `vorta_route_plan` call at the top of
    the transformer forward replaces the per-block Router module calls + top-k + nonzero of the reference.  Buffers
    are allocated once (fixed addresses: a captured hipGraph of the step replays against new routes)."""

    def __init__(self, routers: Sequence[nn.Module]):
        self.routers = list(routers)
        self.heads = self.routers[0].heads
        self.num_experts = self.routers[0].num_experts
        if any(r.heads != self.heads or r.num_experts != self.num_experts for r in self.routers):
            raise ValueError("all routers of a model must have the same number of heads and experts")
        self._key = None
        self._w = self._b = None
        self._out = None
        self._experts_host: Optional[List[List[int]]] = None
        self.tau = 0.0

    def __len__(self):
        return len(self.routers)

    def _stacked(self, device, dtype):
        params = [(r.linear.weight, r.linear.bias) for r in self.routers]
        key = (str(device), dtype) + tuple((w.data_ptr(), w._version, b.data_ptr(), b._version) for w, b in params)
        if key != self._key:  # first use, or a checkpoint was loaded / the model was moved
            self._w = torch.stack([w.detach().to(device=device, dtype=dtype) for w, _ in params]).contiguous()
            self._b = torch.stack([b.detach().to(device=device, dtype=dtype) for _, b in params]).contiguous()
            self._key = key
            self._out = None
        return self._w, self._b

    @torch.no_grad()
    def compute(self, temb: torch.Tensor, tau: float) -> None:
        """temb: (B, E) pure timestep embedding.  Runs the routers in their own dtype (bf16 in the reference's
        scripts, scripts/hunyuan/inference.py:117) -- the scores are rounded exactly like Router.forward."""
        dtype = self.routers[0].linear.weight.dtype
        if dtype not in (torch.bfloat16, torch.float16):
            dtype = torch.bfloat16
        w, b = self._stacked(temb.device, dtype)
        x = temb.detach().to(dtype)
        if self._out is not None and self._out[0].shape[1] != x.shape[0]:
            self._out = None
        if self._out is None:  # first step (or new shapes): allocate the four buffers
            self._out = ops.route_plan(x, w, b, self.heads, float(tau), self.num_experts)
        else:  # the same buffers at the same addresses, through the custom op (traceable; hipGraph replay reads them)
            from .. import torch_ops  # noqa: F401
            torch.ops.vorta.route_plan_(x, w, b, self.heads, float(tau), self.num_experts, *self._out)
        self._experts_host = None
        self.tau = float(tau)

    @property
    def ready(self) -> bool:
        return self._out is not None

    def scores(self, layer: int) -> torch.Tensor:
        return self._out[0][layer]  # (B, H, E)

    def routing(self, layer: int) -> HeadRouting:
        return HeadRouting.from_device(self._out[2][layer], self._out[3][layer])

    def experts_host(self, layer: int) -> List[int]:
        """head -> expert of one layer on the host; ONE device read per step for all layers (the sequence-parallel
        head placement needs it; the reference reads torch.nonzero per expert per layer, hunyuan.py:633)."""
        if self._experts_host is None:
            self._experts_host = self._out[1].cpu().tolist()
        return self._experts_host[layer]


@dataclass
class StepContext:
    """Per-model state shared by the hooks and the bound processors."""
    plan: Optional[RoutePlan] = None
    kwargs: Optional[Dict[str, Any]] = None           # self_attention_kwargs of the running forward
    default_kwargs: Optional[Dict[str, Any]] = None   # set by the pipeline call; used when the forward gets none
    return_routing_scores: bool = False
    default_return_routing_scores: bool = False
    captured: bool = False                            # timestep embedding seen in this forward
    collected: List[Any] = field(default_factory=list)  # per-step score lists gathered for the pipeline call
    step_token: Any = None       # callable -> object identifying the running denoising step (pipeline call)
    last_token: Any = None
    descriptor_cache: Dict[Any, Any] = field(default_factory=dict)
    forwards: int = 0
    needs_tau: bool = True       # hard top-1 dispatch (Eval processors); False for the soft mixture (Train)
    sp_token_shard: bool = False  # pipeline calls: the model gets the WHOLE latent and shards its token sequence itself
    sp_coherent: bool = False     # this pipeline call's ranks were seen to hold the same tokens (install_token_shard)


class BoundProcessor:
    """Adapter between the stock diffusers call `processor(attn, hidden_states, encoder_hidden_states=...,
    attention_mask=..., image_rotary_emb|rotary_emb=...)` and the routed processors' keyword set
    (hunyuan.py:521-539, wan.py:308-328)."""

    def __init__(self, inner, ctx: StepContext, layer: int, rotary_name: str, dense=None):
        self.inner = inner
        self.dense = dense  # plain dense processor serving `use_original_attn=True` (the teacher path)
        self.ctx = ctx
        self.layer = layer
        self.rotary_name = rotary_name
        self._accepted = set(inspect.signature(inner.__call__).parameters)

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, image_rotary_emb=None,
                 rotary_emb=None, use_original_attn: bool = False):
        rotary = image_rotary_emb if self.rotary_name == "image_rotary_emb" else rotary_emb
        ctx = self.ctx
        if use_original_attn:
            if "use_original_attn" in self._accepted:
                return self.inner(attn, hidden_states, encoder_hidden_states, attention_mask, rotary,
                                  use_original_attn=True)
            return self.dense(attn, hidden_states, encoder_hidden_states, attention_mask, rotary)
        if ctx.kwargs is None:
            raise ValueError("this transformer was patched with apply_vorta_transformer: pass `self_attention_kwargs` "
                             "(prepare_*_self_attn_kwargs) to its forward, or run it through vorta_pipeline_call")
        if ctx.plan is None or not ctx.captured:
            raise RuntimeError("routes of this step are not available: the timestep embedder of the model was not "
                               "called before its first attention block")
        kw = {k: v for k, v in ctx.kwargs.items() if k in self._accepted}
        kw["routing_score"] = ctx.plan.scores(self.layer)
        if "head_routing" in self._accepted:
            kw["head_routing"] = ctx.plan.routing(self.layer)
        if "experts_host" in self._accepted and SP_STATE.enabled:
            kw["experts_host"] = ctx.plan.experts_host(self.layer)
        return self.inner(attn, hidden_states, encoder_hidden_states, attention_mask, rotary, **kw)


def set_processor(attn: nn.Module, processor) -> None:
    if hasattr(attn, "set_processor"):
        attn.set_processor(processor)
    else:
        attn.processor = processor


def clear_hooks(model: nn.Module) -> None:
    for h in getattr(model, "_vorta_hooks", []):
        h.remove()
    model._vorta_hooks = []


def add_hook(model: nn.Module, handle) -> None:
    model._vorta_hooks.append(handle)


def context_of(model: nn.Module) -> StepContext:
    ctx = getattr(model, "_vorta_ctx", None)
    if ctx is None:
        ctx = StepContext()
        model._vorta_ctx = ctx
    return ctx


def install_forward_protocol(model: nn.Module, ctx: StepContext, prepare_kwargs=None) -> None:
    """Accept the reference's extra forward keywords on the stock forward and return its output shape:
    `(sample, reg_loss, last_layer_distill_loss, hidden_layer_distill_loss, routing_scores)` when
    `return_dict=False` (modeling_hunyuan.py:441-442, modeling_wan.py:171-172), a RoutedTransformerModelOutput
    otherwise.  `prepare_kwargs(model, args, kwargs, self_attention_kwargs)` may complete the keyword set from
    the forward's own arguments (Hunyuan: the per-prompt sliding-tile descriptor, modeling_hunyuan.py:269-279)."""
    from .outputs import RoutedTransformerModelOutput

    def pre(module, args, kwargs):
        extra = {k: kwargs.pop(k) for k in _EXTRA_FORWARD_KEYS if k in kwargs}
        if extra.get("return_losses") or extra.get("reture_hidden_layer_distill_loss"):
            raise NotImplementedError("router training losses (modeling_hunyuan.py:301-330) are outside the inference "
                                      "hot path of this build")
        sak = extra.get("self_attention_kwargs")
        if sak is None:
            sak = ctx.default_kwargs
        if sak is not None and prepare_kwargs is not None:
            sak = prepare_kwargs(module, args, kwargs, sak)
        ctx.kwargs = sak
        ctx.return_routing_scores = bool(extra.get("return_routing_scores", ctx.default_return_routing_scores))
        ctx.captured = False
        ctx.forwards += 1
        return args, kwargs

    def post(module, args, kwargs, output):
        scores: List[torch.Tensor] = []
        if ctx.return_routing_scores and ctx.plan is not None and ctx.plan.ready:
            # one device read for all layers (the reference does `.detach().cpu()` per block, modeling_hunyuan.py:297)
            all_scores = ctx.plan._out[0].detach().cpu()
            scores = [all_scores[i] for i in range(all_scores.shape[0])]
            # under the pipeline call only the first (conditional) forward of a step is recorded, as the reference
            # passes return_routing_scores=False to the guidance forward (pipeline_hunyuan.py:424-437)
            token = ctx.step_token() if ctx.step_token is not None else None
            if token is None or token is not ctx.last_token:
                ctx.collected.append(scores)
                ctx.last_token = token  # keeps the object alive, so `is` cannot match a recycled id
        ctx.kwargs = None
        if isinstance(output, tuple):
            return (output[0], None, None, None, scores)
        sample = output.sample if hasattr(output, "sample") else output
        return RoutedTransformerModelOutput(sample=sample, routing_scores=scores)

    add_hook(model, model.register_forward_pre_hook(pre, with_kwargs=True))
    add_hook(model, model.register_forward_hook(post, with_kwargs=True))


def install_timestep_capture(embedder: nn.Module, model: nn.Module, ctx: StepContext) -> None:
    """Compute the route plan from the output of the timestep embedder -- the reference's `clean_timesteps_emb`
    (modeling_hunyuan.py:627-628,645) / `temb` before `time_proj` (modeling_wan.py:77) -- at its FIRST call in a
    forward (Hunyuan's token_replace variant embeds a second, zero timestep, modeling_hunyuan.py:633-637)."""

    def hook(module, args, output):
        if ctx.captured or ctx.plan is None or ctx.kwargs is None:
            return None
        tau = ctx.kwargs.get("tau_sparse")
        if tau is None:
            if ctx.needs_tau:
                raise ValueError("self_attention_kwargs has no `tau_sparse` "
                                 "(prepare_*_self_attn_kwargs(..., tau_sparse=...))")
            tau = 0.0  # soft-mixture processors use the scores only; the dispatch lists go unused
        ctx.plan.compute(output, tau)
        ctx.captured = True
        return None

    add_hook(model, embedder.register_forward_hook(hook))


def install_sp_rope(rope: nn.Module, model: nn.Module, frame_dim: int = 2) -> None:
    """Sequence parallelism shards the latent frames (pipeline_hunyuan.py:367-369), but rotary positions are global:
    the reference re-states the rope forward with `num_frames * sp_size` (modeling_hunyuan.py:592-620,
    modeling_wan.py:226-252) and the processors narrow the table to their shard (hunyuan.py:89-95).  The stock rope
    only reads the SHAPE and device of its input, so a zero-stride stand-in of the global shape gives the same
    table without touching the module."""

    ctx = context_of(model)

    def pre(module, args):
        if not SP_STATE.enabled or not args or ctx.sp_token_shard:  # token shard: the input already is the global latent
            return None
        x = args[0]
        shape = list(x.shape)
        shape[frame_dim] *= SP_STATE.sp_size
        ghost = torch.empty((1,), dtype=x.dtype, device=x.device).as_strided(shape, [0] * len(shape))
        return (ghost,) + tuple(args[1:])

    add_hook(model, rope.register_forward_pre_hook(pre))


COHERENCE_SAMPLE = 64


def check_rank_coherence(x: torch.Tensor) -> None:
    """Every sequence-parallel rank must enter the transformer with the SAME hidden states (the ranks keep whole latents and
    cut tokens inside the model): compare a fixed strided SAMPLE of `COHERENCE_SAMPLE` elements across ranks, element by element,
    to 1 % of the sample's largest magnitude.  A checksum would not do: the signed sum and the abs-sum of two
    differently seeded noise tensors differ by ~sqrt(N) against N -- 1e-4 relative at the 5e7..4e8 elements of a Wan / Hunyuan
    hidden state, under any tolerance that lets identical latents through a conv / GEMM whose kernel choice differs between
    processes -- while single elements of identical latents agree to an ulp (<= 0.8 % in bf16) and those of different noise
    differ by their own magnitude, whatever N.  Raises RuntimeError naming the cause (non-finite values are told apart)."""
    flat = x.reshape(-1)
    n = flat.numel()
    idx = torch.linspace(0, n - 1, min(COHERENCE_SAMPLE, n), device=x.device).long()
    sample = flat[idx].float()
    finite = torch.isfinite(x.float().abs().max()).float().reshape(1)
    from ..ulysses import all_gather
    every = all_gather(torch.cat([sample, finite]).reshape(1, -1), dim=0).cpu()  # (P, sample + 1): the one read-back
    if not bool((every[:, -1] > 0).all()) or not bool(torch.isfinite(every).all()):
        raise RuntimeError(
            f"the hidden states entering the transformer are not finite on some rank (finite flags {every[:, -1].tolist()}): "
            "NaN / inf in the latents or the patch embedding, not a rank-coherence problem")
    vals = every[:, :-1]
    tol = 1e-2 * vals.abs().max() + 1e-6
    worst = (vals - vals[:1]).abs().max(dim=1).values
    if not bool((worst <= tol).all()):
        raise RuntimeError(
            "sequence-parallel ranks entered the transformer with different latents (largest difference from rank 0 over "
            f"{vals.shape[1]} sampled elements, per rank: {worst.tolist()}, tolerance {float(tol):.3g}): pass the pipeline a "
            "generator seeded identically on every rank, or none")


def install_token_shard(model: nn.Module, first_block: nn.Module, gather_before: Optional[nn.Module] = None,
                        gather_after: Optional[nn.Module] = None) -> None:
    """Token-level sequence-parallel shard INSIDE the transformer (replaces the frame shard of
    pipeline_hunyuan.py:367-369 for pipeline calls).  The reference shards the latent FRAMES, which refuses every frame
    count P does not divide (129 frames -> 33 latent frames: P = 2, 4, 8 all fail) -- while the token count divides
    (S = 118 800 = 8 x 14 850).  In token mode (`ctx.sp_token_shard`, set by the pipeline call) every rank keeps the
    whole latent; the stock forward embeds it, builds the global rotary table and the global attention mask as in the
    single-process case; the token sequence is cut to this rank's contiguous chunk [r S/P, (r+1) S/P) at the first
    block's input and concatenated again (all-gather along tokens) in front of `gather_before` (the output norm).  The
    blocks in between see exactly what they see under a frame shard: a contiguous S/P chunk + the replicated text.
    `gather_after` instead of `gather_before`: concatenate the OUTPUT of that module (Wan: the last block, whose 16-bit
    output the stock forward widens to fp32 in front of the output norm -- gathering there moves half the bytes).
    Every rank must enter with the same tokens: the first cut of a pipeline call compares a checksum of the sequence
    over the group and refuses to continue on a mismatch (a generator seeded per rank would otherwise make every rank
    denoise a different video outside its own chunk)."""
    from ..ulysses import all_gather
    ctx = context_of(model)
    if (gather_before is None) == (gather_after is None):
        raise ValueError("install_token_shard takes exactly one of gather_before / gather_after")

    def _swap(args, kwargs, fn):
        if "hidden_states" in kwargs:
            kwargs = dict(kwargs, hidden_states=fn(kwargs["hidden_states"]))
            return args, kwargs
        return (fn(args[0]),) + tuple(args[1:]), kwargs

    def cut(module, args, kwargs):
        if not (SP_STATE.enabled and ctx.sp_token_shard):
            return None
        P, r = SP_STATE.sp_size, SP_STATE.group_local_rank

        def f(x):
            S = x.shape[1]
            if S % P:
                raise ValueError(f"{S} video tokens do not divide over {P} sequence-parallel ranks")
            if not ctx.sp_coherent:  # once per pipeline call: one read-back of 65 P floats
                check_rank_coherence(x)
                ctx.sp_coherent = True
            n = S // P
            return x[:, r * n:(r + 1) * n]
        return _swap(args, kwargs, f)

    def join(module, args, kwargs):
        if not (SP_STATE.enabled and ctx.sp_token_shard):
            return None
        return _swap(args, kwargs, lambda x: all_gather(x.contiguous(), dim=1))

    def join_out(module, args, output):
        if not (SP_STATE.enabled and ctx.sp_token_shard):
            return None
        if isinstance(output, tuple):
            return (all_gather(output[0].contiguous(), dim=1),) + tuple(output[1:])
        return all_gather(output.contiguous(), dim=1)

    add_hook(model, first_block.register_forward_pre_hook(cut, with_kwargs=True))
    if gather_before is not None:
        add_hook(model, gather_before.register_forward_pre_hook(join, with_kwargs=True))
    else:
        add_hook(model, gather_after.register_forward_hook(join_out))
