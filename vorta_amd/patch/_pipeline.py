"""Shared body of `sp_pipeline_call` / `vorta_pipeline_call` (vorta/patch/pipeline_hunyuan.py, pipeline_wan.py).

The reference re-states the whole diffusers pipeline `__call__` to change four things: a seed agreed across the
sequence-parallel group (pipeline_hunyuan.py:76-83), a frame shard of the initial latents per rank (:366-369),
`self_attention_kwargs` / `return_routing_scores` passed to every transformer forward (:410-423) and an all-gather
of the latents before decoding (:450-461).  Here the stock `__call__` runs unchanged: the two keyword values reach the
transformer through its step context (vorta_amd/patch/_engine.py); under sequence parallelism the latent stays whole
on every rank and the transformer shards its TOKEN sequence itself (_engine.install_token_shard: any frame count
works, where the reference's frame shard refuses 33 latent frames on 2, 4 or 8 ranks), so the stock call decodes too.
"""
from __future__ import annotations

import inspect
from typing import Any, Callable, Dict, Optional

import torch

from ..ulysses import SP_STATE, all_gather  # noqa: F401  (all_gather: seed agreement)
from . import _engine as E
from .outputs import VideoPipelineOutput

_ORIGINAL_CALL: Dict[type, Callable] = {}
_OURS = set()


def register_pipeline_class(cls: type) -> None:
    """Remember the stock `__call__` of a pipeline class BEFORE a script overwrites it with one of ours
    (`pipeline.__class__.__call__ = vorta_pipeline_call`, scripts/hunyuan/inference.py:104,120).  The diffusers
    classes are registered when this package is imported; call it for any other pipeline class."""
    fn = cls.__call__
    if fn not in _OURS:  # already replaced: the stock call must have been registered earlier
        _ORIGINAL_CALL.setdefault(cls, fn)


def mark_ours(fn: Callable) -> Callable:
    _OURS.add(fn)
    return fn


def original_call(pipe) -> Callable:
    for c in type(pipe).__mro__:
        if c in _ORIGINAL_CALL:
            return _ORIGINAL_CALL[c]
    raise RuntimeError(f"the stock __call__ of {type(pipe).__name__} is unknown: import vorta.patch.pipeline_* (with "
                       f"diffusers installed) or call register_pipeline_class({type(pipe).__name__}) before assigning "
                       f"__call__")


def _shared_generator(pipe) -> torch.Generator:
    """Same seed on every rank of the group -> same initial latents (pipeline_hunyuan.py:76-83)."""
    seed_pt = torch.randint(0, torch.iinfo(torch.int64).max, (1, 1), device=pipe.device, dtype=torch.int64)
    seed = int(all_gather(seed_pt, dim=0)[0].item())
    return torch.Generator(device=pipe.device).manual_seed(seed)


def run(pipe, args, kwargs, *, decode: Callable, self_attention_kwargs: Optional[Dict[str, Any]],
        return_routing_scores: bool, routed: bool):
    orig = original_call(pipe)
    bound = inspect.signature(orig).bind(pipe, *args, **kwargs)
    bound.apply_defaults()
    params = dict(bound.arguments)
    params.pop(next(iter(params)))  # self
    for name, p in inspect.signature(orig).parameters.items():
        if p.kind is inspect.Parameter.VAR_KEYWORD and name in params:
            params.update(params.pop(name))
    output_type, return_dict = params.get("output_type"), params.get("return_dict", True)

    sp = SP_STATE.enabled
    if sp and params.get("generator") is None:
        params["generator"] = _shared_generator(pipe)
    ctx = E.context_of(pipe.transformer)
    if routed:
        ctx.default_kwargs = dict(self_attention_kwargs) if self_attention_kwargs is not None else None
        ctx.default_return_routing_scores = bool(return_routing_scores)
        ctx.collected, ctx.last_token = [], None
        ctx.step_token = lambda: getattr(pipe, "_current_timestep", None)
    # Sequence parallelism: every rank carries the whole latent through the stock loop (same seed -> same latents, the
    # scheduler step is deterministic and tiny) and the transformer shards its TOKEN sequence itself
    # (_engine.install_token_shard).  The reference shards latent frames here (pipeline_hunyuan.py:367-369) and gathers
    # them before decoding (:457-458), which refuses 33 latent frames (129-frame video) on 2, 4 or 8 ranks; the token
    # count of every BASELINE configuration divides by 8.
    ctx.sp_token_shard = bool(sp)
    ctx.sp_coherent = False  # checked at the first cut of this call
    params["return_dict"] = False
    try:
        out = orig(pipe, **params)
    finally:
        ctx.sp_token_shard = False
        if routed:
            ctx.default_kwargs, ctx.default_return_routing_scores, ctx.step_token = None, False, None
    video = out[0]
    scores = ctx.collected if (routed and return_routing_scores) else None
    if routed:
        ctx.collected, ctx.last_token = [], None
    if not return_dict:
        return (video, scores)
    return VideoPipelineOutput(frames=video, routing_scores=scores)
