"""HunyuanVideo transformer patches: `apply_vorta_transformer`, `apply_sp_flashattn_transformer`
(vorta/patch/modeling_hunyuan.py:652-723).

Same entry points, arguments and resulting forward protocol as the reference, which re-states diffusers' transformer,
block, rope and condition-embedding forwards to pass three extra values around.  Here the stock forwards stay and
module hooks carry those values (vorta_amd/patch/_engine.py), so nothing below depends on a diffusers version beyond the
attribute names the reference itself relies on: `transformer_blocks`, `single_transformer_blocks`, `block.attn`
(`heads`, `set_processor`), `block.norm1.linear` / `block.norm.linear` (AdaLN input width = router input width,
modeling_hunyuan.py:674,692), `time_text_embed.timestep_embedder`, `rope`.
"""
import logging
import os
from typing import Any, Dict, Optional

import torch

from ..attention import (HunyuanVideoFlashAttnProcessor, HunyuanVideoFlashAttnProcessorTripleEval,
                         HunyuanVideoFlashAttnProcessorTripleTrain, create_sliding_tile_attn_mask_func)
from . import _engine as E
from .outputs import RoutedTransformerModelOutput  # noqa: F401  (re-exported like the reference module)
from .router import Router, load_router_checkpoint

logger = logging.getLogger(__name__)


def _blocks(model):
    return list(model.transformer_blocks) + list(model.single_transformer_blocks)


def _router_width(block) -> int:
    norm = block.norm1 if hasattr(block, "norm1") else block.norm  # dual / single stream block
    return norm.linear.in_features


def _complete_kwargs(model, args, kwargs, sak: Dict[str, Any]) -> Dict[str, Any]:
    """The sliding-tile descriptor depends on the prompt (padded and valid text length): built here when the
    caller did not (modeling_hunyuan.py:269-279), once per prompt tensor instead of once per forward -- the valid
    length is one host read (`.item()`, as the reference)."""
    if "flex_attn_mask_func" in sak:
        return sak
    names = ("hidden_states", "timestep", "encoder_hidden_states", "encoder_attention_mask")
    mask = kwargs.get("encoder_attention_mask", args[3] if len(args) > 3 else None)
    if mask is None:
        raise ValueError(f"cannot build the sliding-tile descriptor: forward got no `encoder_attention_mask` ({names})")
    ctx = E.context_of(model)
    # One entry, keyed on the mask OBJECT (kept alive by the entry, so its address cannot be handed to the next
    # prompt's mask while we still trust it) and its version counter; the geometry is part of the key too.
    geo = (tuple(mask.shape), tuple(sak["latent_shape"]), tuple(sak["window_size"]), tuple(sak["tile_size"]))
    hit = ctx.descriptor_cache.get("entry")
    if hit is not None and hit[0] is mask and hit[1] == mask._version and hit[2] == geo:
        return dict(sak, flex_attn_mask_func=hit[3])
    desc = create_sliding_tile_attn_mask_func(
        latent_shape=sak["latent_shape"], window_size=sak["window_size"], tile_size=sak["tile_size"],
        text_seq_length=mask.shape[1],
        text_seq_length_no_pad=int(mask.sum(dim=1, dtype=torch.int)[0].item()),  # adhoc: batch_size>1
        device=mask.device)
    ctx.descriptor_cache["entry"] = (mask, mask._version, geo, desc)
    return dict(sak, flex_attn_mask_func=desc)


def apply_vorta_transformer(model, train_router: bool = False, checkpoint_file: Optional[os.PathLike] = None,
                            attn_processor_kwargs: Optional[Dict[str, Any]] = None,
                            router_dtype: Optional[torch.dtype] = None):
    """Mount a Router on every block, install the routed attention processors and the routed forward protocol:
    `model(..., self_attention_kwargs=..., return_routing_scores=...)` -> with `return_dict=False` the 5-tuple
    `(sample, reg_loss, last_layer_distill_loss, hidden_layer_distill_loss, routing_scores)`
    (modeling_hunyuan.py:652-705)."""
    cls = HunyuanVideoFlashAttnProcessorTripleTrain if train_router else HunyuanVideoFlashAttnProcessorTripleEval
    model_dtype = next(model.parameters()).dtype
    router_dtype = router_dtype or model_dtype
    logger.info(f"Model {model.__class__.__name__} ({model_dtype=}) is mounted with Router ({router_dtype=})")

    ctx = E.context_of(model)
    ctx.needs_tau = not train_router
    E.clear_hooks(model)
    kw = dict(attn_processor_kwargs or {})
    kw.update(check_input=True)
    dense = HunyuanVideoFlashAttnProcessor()
    blocks = _blocks(model)
    for layer, block in enumerate(blocks):
        if not hasattr(block, "router"):
            ref = next(block.parameters())
            block.router = Router(embedding_dim=_router_width(block), heads=block.attn.heads).to(
                device=ref.device, dtype=router_dtype)
        if train_router:
            block.router.requires_grad_(True)
        E.set_processor(block.attn, E.BoundProcessor(cls(**kw), ctx, layer, "image_rotary_emb", dense=dense))
        kw.update(check_input=False)  # only the first block checks its input (modeling_hunyuan.py:684)
    ctx.plan = E.RoutePlan([b.router for b in blocks])

    E.install_forward_protocol(model, ctx, _complete_kwargs)
    E.install_timestep_capture(model.time_text_embed.timestep_embedder, model, ctx)
    E.install_sp_rope(model.rope, model)
    E.install_token_shard(model, model.transformer_blocks[0], model.norm_out)
    if checkpoint_file is not None:
        load_router_checkpoint(checkpoint_file, model)
    return model


def apply_sp_flashattn_transformer(model):
    """`--native_attention`: dense attention for every head through the HIP kernel, sequence-parallel aware
    (modeling_hunyuan.py:708-723)."""
    E.context_of(model)
    E.clear_hooks(model)
    for block in _blocks(model):
        E.set_processor(block.attn, HunyuanVideoFlashAttnProcessor())
    E.install_sp_rope(model.rope, model)
    E.install_token_shard(model, model.transformer_blocks[0], model.norm_out)
    return model
