"""`torch.ops.vorta.*`: the launchers of vorta_amd/ops.py registered as PyTorch custom ops.

The C ABI (include/vorta_hip.h) is the boundary; ops.py binds it with ctypes.  This module puts the same launchers
behind `torch.library.custom_op`, so code that is traced (`torch.compile`, export) sees them as opaque, stream-correct
operators with declared mutations and shape functions instead of Python it cannot follow (BASELINE.json north star:
"Python host code calls into hand-written HIP kernels through PyTorch-ROCm custom ops (thin C-ABI ...)").  The
attention processors of vorta_amd/attention call THESE operators (`vorta::routed_attention`, `vorta::attn_fwd`,
`vorta::qk_norm_rope`, `vorta::route_scores`, `vorta::soft_mixture_attention`; the route plan of patch/_engine.py
`vorta::route_plan_`): a processor `__call__` traces under `torch.compile(fullgraph=True)` with no graph break, and the
eager path runs the very same launchers behind one dispatcher hop per operator.

Registered (tensors are (H,S,D) views as in ops.py; optional tensors may be None):
    vorta::attn_fwd(q,k,v,out, n_q,n_kv, ...)         -> ()    mutates out        vorta_attn_fwd
    vorta::coreset_select(x, latent, group, n_keep, ...) -> (keep_rows, drop_rows)  vorta_coreset_select
    vorta::sta_build_tables(like, latent, tile, window, t_eff, row_map) -> (q_rows, kv_rows)
    vorta::route_scores(scores, tau)                  -> (expert_of_head, head_lists, head_counts)
    vorta::router_route(temb, weight, bias, heads, tau) -> (scores, expert_of_head, head_lists, head_counts)
    vorta::qk_norm_rope(x, weight, eps, cos, sin, rope_tokens, across_heads) -> () mutates x
    vorta::mix_experts(x0,x1,x2, scores, out)         -> ()    mutates out
    vorta::routed_attention(q,k,v,out, head_lists, head_counts, counts_host, latent, tile, window, group, rate, model, ...)
                                                      -> ()    mutates out        the per-layer routed op (routed.py)
    vorta::soft_mixture_attention(q,k,v, scores, out, latent, ...) -> () mutates out   the training-time forward
    vorta::route_plan_(temb, weight, bias, heads, tau, n_experts, scores, expert, lists, counts) -> () mutates the four
"""
from typing import List, Optional, Tuple

import torch

from . import ops


@torch.library.custom_op("vorta::attn_fwd", mutates_args=("out",), device_types="cuda")
def attn_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, n_q: int, n_kv: int,
             head_list: Optional[torch.Tensor] = None, n_heads: int = -1, n_heads_dev: Optional[torch.Tensor] = None,
             q_group_len: int = 0, q_row_offset: int = 0, q_valid: int = -1, q_rows: Optional[torch.Tensor] = None,
             kv_row_offset: int = 0, kv_rows: Optional[torch.Tensor] = None, kv_rows_stride_g: int = 0,
             dup_rows: Optional[torch.Tensor] = None, n_dup_pos: int = 0, scale: float = 0.0, block_rows: int = 0,
             n_splits: int = 1, n_kv_dev: Optional[torch.Tensor] = None, q_valid_dev: Optional[torch.Tensor] = None,
             variant: int = 0) -> None:
    ops.attn_fwd(q, k, v, out, n_q=n_q, n_kv=n_kv, head_list=head_list, n_heads=None if n_heads < 0 else n_heads,
                 n_heads_dev=n_heads_dev, q_group_len=q_group_len, q_row_offset=q_row_offset,
                 q_valid=None if q_valid < 0 else q_valid, q_rows=q_rows, kv_row_offset=kv_row_offset, kv_rows=kv_rows,
                 kv_rows_stride_g=kv_rows_stride_g, dup_rows=dup_rows, n_dup_pos=n_dup_pos,
                 scale=None if scale <= 0.0 else scale, block_rows=block_rows, n_splits=n_splits, n_kv_dev=n_kv_dev,
                 q_valid_dev=q_valid_dev, variant=variant)


@attn_fwd.register_fake
def _(q, k, v, out, n_q, n_kv, head_list=None, n_heads=-1, n_heads_dev=None, q_group_len=0, q_row_offset=0, q_valid=-1,
      q_rows=None, kv_row_offset=0, kv_rows=None, kv_rows_stride_g=0, dup_rows=None, n_dup_pos=0, scale=0.0,
      block_rows=0, n_splits=1, n_kv_dev=None, q_valid_dev=None, variant=0) -> None:
    return None


def _coreset_shapes(x, latent, group, n_keep, head_list, n_heads, n_tail):
    slots = n_heads if n_heads >= 0 else (head_list.numel() if head_list is not None else x.shape[0])
    g = group[0] * group[1] * group[2]
    G = (latent[0] // group[0]) * (latent[1] // group[1]) * (latent[2] // group[2])
    return (slots, G * (1 + n_keep) + n_tail), (slots, G, g - 1 - n_keep)


@torch.library.custom_op("vorta::coreset_select", mutates_args=(), device_types="cuda")
def coreset_select(x: torch.Tensor, latent: List[int], group: List[int], n_keep: int,
                   head_list: Optional[torch.Tensor] = None, n_heads: int = -1,
                   n_heads_dev: Optional[torch.Tensor] = None, tail_first: int = 0, n_tail: int = 0,
                   row_map: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    keep, drop = ops.coreset_select(x, latent, group, n_keep, head_list=head_list,
                                    n_heads=None if n_heads < 0 else n_heads, n_heads_dev=n_heads_dev,
                                    tail_first=tail_first, n_tail=n_tail, row_map=row_map)
    return keep, drop


@coreset_select.register_fake
def _(x, latent, group, n_keep, head_list=None, n_heads=-1, n_heads_dev=None, tail_first=0, n_tail=0, row_map=None):
    ks, ds = _coreset_shapes(x, latent, group, n_keep, head_list, n_heads, n_tail)
    return x.new_empty(ks, dtype=torch.int32), x.new_empty(ds, dtype=torch.int32)


@torch.library.custom_op("vorta::sta_build_tables", mutates_args=(), device_types="cuda")
def sta_build_tables(like: torch.Tensor, latent: List[int], tile: List[int], window: List[int], t_eff: int = 0,
                     row_map: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """`like` only names the device (custom ops take no device argument)."""
    q_rows, kv_rows = ops.sta_build_tables(latent, tile, window, t_eff, like.device, row_map=row_map)
    return q_rows, kv_rows


@sta_build_tables.register_fake
def _(like, latent, tile, window, t_eff=0, row_map=None):
    n_tiles, _, n_kv = ops.sta_table_sizes(latent, tile, window, t_eff)  # host-only geometry query
    S = latent[0] * latent[1] * latent[2]
    return like.new_empty((S,), dtype=torch.int32), like.new_empty((n_tiles, n_kv), dtype=torch.int32)


@torch.library.custom_op("vorta::route_scores", mutates_args=(), device_types="cuda")
def route_scores(scores: torch.Tensor, tau: float) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    return ops.route_scores(scores, tau)


@route_scores.register_fake
def _(scores, tau):
    _, H, E = scores.shape
    i32 = dict(dtype=torch.int32)
    return scores.new_empty((H,), **i32), scores.new_empty((E, H), **i32), scores.new_empty((E,), **i32)


@torch.library.custom_op("vorta::router_route", mutates_args=(), device_types="cuda")
def router_route(temb: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, heads: int, tau: float,
                 n_experts: int = 3) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    return ops.router_route(temb, weight, bias, heads, tau, n_experts)


@router_route.register_fake
def _(temb, weight, bias, heads, tau, n_experts=3):
    i32 = dict(dtype=torch.int32)
    return (temb.new_empty((temb.shape[0], heads, n_experts)), temb.new_empty((heads,), **i32),
            temb.new_empty((n_experts, heads), **i32), temb.new_empty((n_experts,), **i32))


@torch.library.custom_op("vorta::qk_norm_rope", mutates_args=("x",), device_types="cuda")
def qk_norm_rope(x: torch.Tensor, weight: Optional[torch.Tensor], eps: float, cos: Optional[torch.Tensor] = None,
                 sin: Optional[torch.Tensor] = None, rope_tokens: int = 0, across_heads: bool = False) -> None:
    """rope_tokens < 0: every token the cos / sin tables cover (ops.qk_norm_rope's default)"""
    ops.qk_norm_rope(x, weight, eps, cos=cos, sin=sin, rope_tokens=None if rope_tokens < 0 else rope_tokens,
                     across_heads=across_heads)


@qk_norm_rope.register_fake
def _(x, weight, eps, cos=None, sin=None, rope_tokens=0, across_heads=False) -> None:
    return None


@torch.library.custom_op("vorta::mix_experts", mutates_args=("out",), device_types="cuda")
def mix_experts(x0: torch.Tensor, x1: torch.Tensor, x2: torch.Tensor, scores: torch.Tensor, out: torch.Tensor) -> None:
    ops.mix_experts([x0, x1, x2], scores, out)


@mix_experts.register_fake
def _(x0, x1, x2, scores, out) -> None:
    return None


@torch.library.custom_op("vorta::routed_attention", mutates_args=("out",), device_types="cuda")
def routed_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, head_lists: torch.Tensor,
                     head_counts: Optional[torch.Tensor], counts_host: Optional[List[int]], latent: List[int],
                     tile: List[int], window: List[int], group: List[int], rate: float, model: str, text_len: int = 0,
                     text_valid: int = 0, scale: float = 0.0, precision: str = "",
                     row_map: Optional[torch.Tensor] = None) -> None:
    """The routed sparse-attention op of one layer (vorta_amd/routed.py: hunyuan.py:556-605 / wan.py:351-383): q,k,v,out
    (1,H,N,D) views; head_lists (3,H) int32 + head_counts (3,) int32 on the device (the route plan's / vorta::route_scores'
    output: no host read) or `counts_host` = the three counts known on the host; the geometry by value (its tables are
    cached per process, routed.geometry_for).  precision "" = the process default (set_attention_precision)."""
    from . import routed
    geom = routed.geometry_for(latent, tile, window, group, rate, q.device, row_map=row_map)
    routing = routed.HeadRouting(head_lists, list(counts_host) if counts_host is not None else None, head_counts)
    fp8 = None if precision == "" else (False if precision == "native" else True if precision == "fp8" else precision)
    routed.routed_attention(q, k, v, routing, geom, model=model, text_len=text_len, text_valid=text_valid, out=out,
                            scale=None if scale <= 0.0 else scale, fp8=fp8)


@routed_attention.register_fake
def _(q, k, v, out, head_lists, head_counts, counts_host, latent, tile, window, group, rate, model, text_len=0,
      text_valid=0, scale=0.0, precision="", row_map=None) -> None:
    return None


@torch.library.custom_op("vorta::soft_mixture_attention", mutates_args=("out",), device_types="cuda")
def soft_mixture_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, routing_score: torch.Tensor, out: torch.Tensor,
                           latent: List[int], tile: List[int], window: List[int], group: List[int], rate: float, model: str,
                           text_len: int = 0, text_valid: int = 0, scale: float = 0.0) -> None:
    """hunyuan.py:375-408,509-513 / wan.py:226-241,296-300, forward only (routed.soft_mixture_attention)"""
    from . import routed
    geom = routed.geometry_for(latent, tile, window, group, rate, q.device)
    routed.soft_mixture_attention(q, k, v, routing_score, geom, model=model, text_len=text_len, text_valid=text_valid,
                                  out=out, scale=None if scale <= 0.0 else scale)


@soft_mixture_attention.register_fake
def _(q, k, v, routing_score, out, latent, tile, window, group, rate, model, text_len=0, text_valid=0, scale=0.0) -> None:
    return None


@torch.library.custom_op("vorta::route_plan_", mutates_args=("scores", "expert", "lists", "counts"), device_types="cuda")
def route_plan_(temb: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, heads: int, tau: float, n_experts: int,
                scores: torch.Tensor, expert: torch.Tensor, lists: torch.Tensor, counts: torch.Tensor) -> None:
    """vorta_route_plan into caller-owned buffers at fixed addresses (patch/_engine.py RoutePlan; hipGraph replay)"""
    ops.route_plan(temb, weight, bias, heads, tau, n_experts, out=(scores, expert, lists, counts))


@route_plan_.register_fake
def _(temb, weight, bias, heads, tau, n_experts, scores, expert, lists, counts) -> None:
    return None


def geometry_args(lowres_group_info, window_size, tile_size, latent_shape) -> dict:
    """the geometry keywords of vorta::routed_attention / vorta::soft_mixture_attention from a processor's keywords"""
    return dict(latent=[int(x) for x in latent_shape], tile=[int(x) for x in tile_size], window=[int(x) for x in window_size],
                group=[int(x) for x in lowres_group_info.window_size], rate=float(lowres_group_info.reduction_rate))


def routing_args(head_routing) -> dict:
    """head_lists / head_counts / counts_host of a routed.HeadRouting"""
    return dict(head_lists=head_routing.lists, head_counts=head_routing.counts_dev,
                counts_host=None if head_routing.counts_host is None else [int(c) for c in head_routing.counts_host])
