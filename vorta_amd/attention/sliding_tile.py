"""Sliding-tile geometry descriptor: what this build passes under the reference's `flex_attn_mask_func`
keyword (vorta/attention/sliding_attn_flex.py:72-134 returns a FlexAttention BlockMask there; the argument is
opaque to every caller)."""
from dataclasses import dataclass, field
from typing import Optional, Tuple

import torch


@dataclass
class SlidingTileDescriptor:
    latent_shape: Tuple[int, int, int]
    window_size: Tuple[int, int, int]
    tile_size: Tuple[int, int, int]
    text_seq_length: int
    text_seq_length_no_pad: int
    device: torch.device
    _tables: Optional[tuple] = field(default=None, repr=False)

    def tables(self, row_map: Optional[torch.Tensor] = None):
        """(q_rows, kv_rows, n_kv) built by vorta_sta_build_tables on first use."""
        from .. import ops
        if row_map is not None:
            q, kv = ops.sta_build_tables(self.latent_shape, self.tile_size, self.window_size,
                                         self.text_seq_length_no_pad, self.device, row_map=row_map)
            return q, kv, kv.shape[1]
        if self._tables is None:
            q, kv = ops.sta_build_tables(self.latent_shape, self.tile_size, self.window_size,
                                         self.text_seq_length_no_pad, self.device)
            self._tables = (q, kv, kv.shape[1])
        return self._tables


def create_sliding_tile_attn_mask_func(latent_shape, window_size, tile_size, text_seq_length: int,
                                       text_seq_length_no_pad: int, device) -> SlidingTileDescriptor:
    """Same signature as sliding_attn_flex.py:72-79.  Validates what the reference validates later
    (hunyuan.py:264-267) and returns the descriptor; the tables are built lazily on the device."""
    latent_shape, window_size, tile_size = (tuple(int(v) for v in x) for x in (latent_shape, window_size, tile_size))
    for t, l in zip(tile_size, latent_shape):
        if l % t != 0:
            raise ValueError(f"Tile size {tile_size} (dim={t}) does not divide latent shape {latent_shape} (dim={l}).")
    return SlidingTileDescriptor(latent_shape, window_size, tile_size, int(text_seq_length),
                                 int(text_seq_length_no_pad), torch.device(device))
