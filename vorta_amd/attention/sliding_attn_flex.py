"""`vorta.attention.sliding_attn_flex` by name: `create_sliding_tile_attn_mask_func` and `sliding_tile_flex_attn`
(vorta/attention/sliding_attn_flex.py:72-211), MI355X-native.

The reference tiles Q/K/V (4 permute passes), concatenates the text, runs a compiled FlexAttention closure over a
BlockMask and un-tiles the output.  Here the descriptor returned by `create_sliding_tile_attn_mask_func` stands in for
that closure and the call is ONE launch of the gather attention kernel (+1 for Hunyuan's text queries) reading the
raster-order tensors through the tile tables; there is no tiled copy of anything.
"""
from typing import Optional, Tuple, Union

import torch

from ..routed import HeadRouting, geometry_for, routed_attention
from ..ulysses import SP_STATE
from .sliding_tile import SlidingTileDescriptor, create_sliding_tile_attn_mask_func  # noqa: F401


def sliding_tile_flex_attn(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                           flex_attn_func: SlidingTileDescriptor,
                           encoder_query: Optional[torch.Tensor] = None, encoder_key: Optional[torch.Tensor] = None,
                           encoder_value: Optional[torch.Tensor] = None,
                           tile_size: Tuple[int, int, int] = (6, 8, 8), latent_shape: Tuple[int, int, int] = (30, 45, 80),
                           head_dim: int = 2) -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
    """Sliding-tile attention of every head.  Arguments as sliding_attn_flex.py:137-148: tensors are (B,S,H,D) when
    `head_dim == 2`, (B,H,S,D) when `head_dim == 1`; with `encoder_*` (MMDiT) the text is appended and the pair
    (video, text) is returned.  `flex_attn_func` is the descriptor made by `create_sliding_tile_attn_mask_func`."""
    if SP_STATE.enabled:
        raise NotImplementedError("under sequence parallelism the processors reshard all heads once "
                                  "(vorta_amd/attention/_sp.py); this helper is the single-GPU entry point")
    if not isinstance(flex_attn_func, SlidingTileDescriptor):
        raise TypeError("flex_attn_func must be the descriptor returned by create_sliding_tile_attn_mask_func")
    desc = flex_attn_func
    if tuple(tile_size) != desc.tile_size or tuple(latent_shape) != desc.latent_shape:
        raise ValueError(f"tile_size {tuple(tile_size)} / latent_shape {tuple(latent_shape)} do not match the descriptor "
                         f"({desc.tile_size}, {desc.latent_shape})")
    mmdit = encoder_query is not None
    if head_dim == 2:
        query, key, value = (x.transpose(1, 2) for x in (query, key, value))
        if mmdit:
            encoder_query, encoder_key, encoder_value = (x.transpose(1, 2) for x in (encoder_query, encoder_key, encoder_value))
    elif head_dim != 1:
        raise ValueError("head_dim must be 1 (B,H,S,D) or 2 (B,S,H,D)")
    if query.shape[0] != 1:
        raise AssertionError(f"Batch size {query.shape[0]} is not supported by sliding_tile_flex_attn.")
    T = 0
    if mmdit:
        T = encoder_query.shape[2]
        if T != desc.text_seq_length:
            raise ValueError(f"text length {T} does not match the descriptor ({desc.text_seq_length})")
        query, key, value = (torch.cat([a, b], dim=2) for a, b in ((query, encoder_query), (key, encoder_key),
                                                                  (value, encoder_value)))
    H = query.shape[1]
    geom = geometry_for(desc.latent_shape, desc.tile_size, desc.window_size, (1, 1, 1), 0.0, query.device)
    buf = torch.empty((1, query.shape[2], H, query.shape[3]), dtype=query.dtype, device=query.device)
    routed_attention(query, key, value, HeadRouting.from_expert_ids([2] * H, query.device), geom,
                     model="hunyuan" if mmdit else "wan", text_len=T, text_valid=desc.text_seq_length_no_pad if mmdit else 0,
                     out=buf.permute(0, 2, 1, 3))
    out = buf if head_dim == 2 else buf.permute(0, 2, 1, 3)  # (B,S,H,D) or (B,H,S,D)
    if not mmdit:
        return out
    S = query.shape[2] - T
    return (out[:, :S], out[:, S:]) if head_dim == 2 else (out[:, :, :S], out[:, :, S:])
