"""`vorta.attention.tile` by name: raster order <-> tile-major order of a video token sequence
(vorta/attention/tile.py:7-78).  Compatibility helpers: the kernels of this build never materialise the tiled
layout (they read through row tables, vorta_sta_build_tables); these are one index_select each."""
from typing import Tuple

import torch


def _tile_major_index(sp_size: int, tile_size: Tuple[int, int, int], latent_shape: Tuple[int, int, int], device):
    """position in the tiled sequence -> position in the input sequence.  The input is the rank-major concatenation of
    `sp_size` frame chunks, each in (t, h, w) raster order (`(sp t h w)`, tile.py:21-25)."""
    t, h, w = latent_shape
    tt, th, tw = tile_size
    if t % tt or h % th or w % tw or t % sp_size:
        raise ValueError(f"tile {tile_size} / sp_size {sp_size} do not divide latent shape {latent_shape}")
    tl = t // sp_size
    src = torch.arange(t * h * w, device=device).view(sp_size, tl, h, w)
    # `(sp t h w) -> (t sp h w)`: the reference interleaves the ranks' frames (identity for sp_size 1)
    x = src.permute(1, 0, 2, 3).reshape(t, h, w)
    x = x.view(t // tt, tt, h // th, th, w // tw, tw).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(-1)


def tile_layout(x: torch.Tensor, sp_size: int, tile_size: Tuple[int, int, int], latent_shape: Tuple[int, int, int],
                head_dim: int = 2) -> torch.Tensor:
    """x: (B,H,S,D) if head_dim == 1, (B,S,H,D) if head_dim == 2 (the reference's default); returns the same shape
    with S in tile-major order."""
    idx = _tile_major_index(sp_size, tuple(tile_size), tuple(latent_shape), x.device)
    return x.index_select(2 if head_dim == 1 else 1, idx)


def untile_layout(x: torch.Tensor, sp_size: int, tile_size: Tuple[int, int, int], latent_shape: Tuple[int, int, int],
                  head_dim: int = 2) -> torch.Tensor:
    """inverse of tile_layout (tile.py:44-78)."""
    idx = _tile_major_index(sp_size, tuple(tile_size), tuple(latent_shape), x.device)
    inv = torch.empty_like(idx)
    inv[idx] = torch.arange(idx.numel(), device=x.device)
    return x.index_select(2 if head_dim == 1 else 1, inv)
