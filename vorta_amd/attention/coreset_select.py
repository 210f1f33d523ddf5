"""Coreset ("low-res") token selection: public surface of vorta/attention/coreset_select.py.

On the hot path none of the tensors below is materialised: `vorta_coreset_select` (HIP) writes keep/drop row
lists and the attention kernel gathers / scatters through them (vorta_amd/routed.py).  `get_group_info`
keeps the reference's index tables for API compatibility and carries the geometry the kernels need;
`pool_sequence_by_similarity` / `unpool_sequence_by_similarity` are compatibility helpers (GPU only, the
ranking comes from the HIP kernel, the data movement is a torch index op) for code that calls them directly.
"""
from dataclasses import dataclass
from typing import Optional, Tuple

import torch


@dataclass
class LowresGroupInfo:
    """coreset_select.py:8-12, plus the closed-form geometry the HIP kernels use instead of the tables."""
    center_indices: torch.Tensor  # (G, 1) int64
    margin_indices: torch.Tensor  # (G, g-1) int64
    num_unpooled_tokens_per_group: int
    latent_shape: Tuple[int, int, int] = (0, 0, 0)
    window_size: Tuple[int, int, int] = (0, 0, 0)
    reduction_rate: float = 0.5


def get_group_info(latent_video_shape: Tuple[int, int, int], compress_window_size: Tuple[int, int, int],
                   reduction_rate: float = 0.5, device: torch.device = torch.device("cpu")) -> LowresGroupInfo:
    """Window groups of the (frame, height, width) token grid (coreset_select.py:15-60): row g lists the raster
    ids of window g in in-window raster order, split into the centre token and the g-1 margins."""
    f, h, w = (int(x) for x in latent_video_shape)
    fw, hw, ww = (int(x) for x in compress_window_size)
    nf, nh, nw = f // fw, h // hw, w // ww
    ids = torch.arange(f * h * w).view(f, h, w)[: nf * fw, : nh * hw, : nw * ww]  # partial windows are cropped
    groups = ids.view(nf, fw, nh, hw, nw, ww).permute(0, 2, 4, 1, 3, 5).reshape(nf * nh * nw, fw * hw * ww).to(device)
    c = (fw // 2) * hw * ww + (hw // 2) * ww + ww // 2
    keep = torch.ones(fw * hw * ww, dtype=torch.bool, device=groups.device)
    keep[c] = False
    return LowresGroupInfo(center_indices=groups[:, c:c + 1], margin_indices=groups[:, keep],
                           num_unpooled_tokens_per_group=int(fw * hw * ww * (1 - reduction_rate)) - 1,
                           latent_shape=(f, h, w), window_size=(fw, hw, ww), reduction_rate=reduction_rate)


@dataclass
class MatchingResults:
    """Opaque to callers, like the reference's (coreset_select.py:62-65); here it holds raster row lists."""
    keep_rows: torch.Tensor  # (B*H, G*(1+n_keep)) int32: packed-sequence position -> token id
    drop_rows: torch.Tensor  # (B*H, G, g-1-n_keep) int32: dropped margins of each group


def pool_sequence_by_similarity(hidden_states: torch.Tensor, lowres_group_info: LowresGroupInfo,
                                matching_results: Optional[MatchingResults] = None
                                ) -> Tuple[torch.Tensor, MatchingResults]:
    """(B,H,S,D) -> packed (B,H,G*(1+n_keep),D) = [centres | kept margins, least similar first]
    (coreset_select.py:68-124).  Compatibility helper; the routed op never calls it."""
    from .. import ops
    B, H, S, D = hidden_states.shape
    if matching_results is None:
        x = hidden_states.contiguous().view(B * H, S, D)
        keep, drop = ops.coreset_select(x, lowres_group_info.latent_shape, lowres_group_info.window_size,
                                        lowres_group_info.num_unpooled_tokens_per_group)
        matching_results = MatchingResults(keep, drop)
    idx = matching_results.keep_rows.view(B, H, -1, 1).long().expand(-1, -1, -1, D)
    return torch.gather(hidden_states, 2, idx), matching_results


def unpool_sequence_by_similarity(pooled_hidden_states: torch.Tensor, lowres_group_info: LowresGroupInfo,
                                  matching_results: MatchingResults) -> torch.Tensor:
    """inverse scatter; dropped margins receive their centre's row (coreset_select.py:127-185)."""
    B, H, _, D = pooled_hidden_states.shape
    G = lowres_group_info.center_indices.shape[0]
    g = 1 + lowres_group_info.margin_indices.shape[1]
    out = torch.zeros((B, H, G * g, D), dtype=pooled_hidden_states.dtype, device=pooled_hidden_states.device)
    keep = matching_results.keep_rows.view(B, H, -1, 1).long().expand(-1, -1, -1, D)
    out.scatter_(2, keep, pooled_hidden_states)
    drop = matching_results.drop_rows.view(B, H, G, -1)
    n_drop = drop.shape[-1]
    centres = pooled_hidden_states[:, :, :G].repeat_interleave(n_drop, dim=2)
    out.scatter_(2, drop.reshape(B, H, -1, 1).long().expand(-1, -1, -1, D), centres)
    return out
