"""Wan transformer patches: `apply_vorta_transformer`, `apply_sp_flashattn_transformer`
(vorta/patch/modeling_wan.py:255-323).  See modeling_hunyuan.py / _engine.py for the mechanism.

Attribute names relied on (the reference's own, modeling_wan.py:273-304): `blocks`, `block.attn1` (self attention),
`block.attn2` (cross attention), `condition_embedder.time_embedder` / `.time_proj`, `rope`.
"""
import logging
import os
from typing import Any, Dict, Optional

import torch

from ..attention import WanAttnProcessor2_0, WanAttnProcessorTripleEval, WanAttnProcessorTripleTrain
from . import _engine as E
from .outputs import RoutedTransformerModelOutput  # noqa: F401
from .router import Router, load_router_checkpoint

logger = logging.getLogger(__name__)


def apply_vorta_transformer(model, train_router: bool = False, checkpoint_file: Optional[os.PathLike] = None,
                            attn_processor_kwargs: Optional[Dict[str, Any]] = None,
                            router_dtype: Optional[torch.dtype] = None):
    """Router per block on the timestep embedding BEFORE `time_proj` (modeling_wan.py:77,127,284), routed processor
    on `attn1`, sequence-parallel dense processor on `attn2` (modeling_wan.py:287-304)."""
    cls = WanAttnProcessorTripleTrain if train_router else WanAttnProcessorTripleEval
    dtype = router_dtype or next(model.parameters()).dtype
    logger.info(f"Model {model.__class__.__name__} is mounted with Router({dtype=})")
    embedding_dim = model.condition_embedder.time_proj.in_features

    ctx = E.context_of(model)
    ctx.needs_tau = not train_router
    E.clear_hooks(model)
    kw = dict(attn_processor_kwargs or {})
    kw.update(check_input=True)
    blocks = list(model.blocks)
    for layer, block in enumerate(blocks):
        if not hasattr(block, "router"):
            ref = next(block.parameters())
            block.router = Router(embedding_dim=embedding_dim, heads=block.attn1.heads, num_experts=3).to(
                device=ref.device, dtype=dtype)
        if train_router:
            block.router.requires_grad_(True)
        E.set_processor(block.attn1, E.BoundProcessor(cls(**kw), ctx, layer, "rotary_emb"))
        E.set_processor(block.attn2, WanAttnProcessor2_0())
        kw.update(check_input=False)
    ctx.plan = E.RoutePlan([b.router for b in blocks])

    E.install_forward_protocol(model, ctx)
    E.install_timestep_capture(model.condition_embedder.time_embedder, model, ctx)
    E.install_sp_rope(model.rope, model)
    # gather the last block's 16-bit output: the stock forward hands norm_out an fp32 copy (twice the bytes)
    E.install_token_shard(model, model.blocks[0], gather_after=model.blocks[-1])
    if checkpoint_file is not None:
        load_router_checkpoint(checkpoint_file, model)
    return model


def apply_sp_flashattn_transformer(model):
    """`--native_attention` (modeling_wan.py:310-323)."""
    E.context_of(model)
    E.clear_hooks(model)
    for block in model.blocks:
        E.set_processor(block.attn1, WanAttnProcessor2_0())
        E.set_processor(block.attn2, WanAttnProcessor2_0())
    E.install_sp_rope(model.rope, model)
    # gather the last block's 16-bit output: the stock forward hands norm_out an fp32 copy (twice the bytes)
    E.install_token_shard(model, model.blocks[0], gather_after=model.blocks[-1])
    return model
