"""Keyword preparation for the routed processors and the pixel -> token grid map (vorta/patch/utils.py)."""
from typing import Any, Dict, Optional, Tuple

import torch

from ..attention import create_sliding_tile_attn_mask_func, get_group_info


def validate_geometry(kw: Dict[str, Any]) -> None:
    """Fail at configuration time, with the messages the processors raise at the first forward
    (hunyuan.py:260-272), when the `self_attention_kwargs` of a checkpoint's config.json do not fit the latent grid
    of the requested video size: the published (6,9,8)/(2,3,2) geometry fits 117 frames, not 129."""
    latent = tuple(int(v) for v in kw["latent_shape"])
    tile = tuple(int(v) for v in kw.get("tile_size", (1, 1, 1)))
    for t_size, l_size in zip(tile, latent):
        if l_size % t_size != 0:
            raise ValueError(f"Tile size {tile} (dim={t_size}) does not divide latent shape {latent} (dim={l_size}).")
    group = kw.get("lowres_window_size")
    if group is not None:
        for g_size, l_size in zip(group, latent):
            if l_size % g_size != 0:
                raise ValueError(f"Low-res window {tuple(group)} (dim={g_size}) does not divide latent shape {latent} "
                                 f"(dim={l_size}): the coreset expert would crop the sequence.")


def _add_group_info(kw: Dict[str, Any], device) -> None:
    validate_geometry(kw)
    kw.update(lowres_group_info=get_group_info(kw["latent_shape"], kw.pop("lowres_window_size"),
                                               reduction_rate=kw.pop("lowres_reduction_rate"), device=device))


def prepare_wan_self_attn_kwargs(self_attention_kwargs: Dict[str, Any], device: torch.device,
                                 tau_sparse: Optional[float] = None) -> Dict[str, Any]:
    """vorta/patch/utils.py:8-36: pops lowres_window_size / lowres_reduction_rate, adds lowres_group_info, the
    (text-free) sliding-tile descriptor under `flex_attn_mask_func`, and tau_sparse."""
    _add_group_info(self_attention_kwargs, device)
    self_attention_kwargs.update(flex_attn_mask_func=create_sliding_tile_attn_mask_func(
        latent_shape=self_attention_kwargs["latent_shape"], window_size=self_attention_kwargs["window_size"],
        tile_size=self_attention_kwargs["tile_size"], text_seq_length=0, text_seq_length_no_pad=0, device=device))
    if tau_sparse is not None:
        self_attention_kwargs.update(tau_sparse=tau_sparse)
    return self_attention_kwargs


def prepare_hunyuan_self_attn_kwargs(self_attention_kwargs: Dict[str, Any], device: torch.device,
                                     tau_sparse: Optional[float] = None) -> Dict[str, Any]:
    """vorta/patch/utils.py:39-56: the sliding-tile descriptor depends on the prompt's text length and is added
    per prompt by the pipeline call (pipeline_hunyuan.py:378-392)."""
    _add_group_info(self_attention_kwargs, device)
    if tau_sparse is not None:
        self_attention_kwargs.update(tau_sparse=tau_sparse)
    return self_attention_kwargs


class Pixel2TokenFactory:
    """video (frames, height, width) in pixels -> latent token grid (vorta/patch/utils.py:59-92)."""

    def __init__(self, temporal_vae: int, spatial_vae: int, temporal_patchfy: int = 1, spatial_patchfy: int = 2):
        self.temporal_total = temporal_vae * temporal_patchfy
        self.spatial_total = spatial_vae * spatial_patchfy

    def __call__(self, video_shape: Tuple[int, int, int]) -> Tuple[int, int, int]:
        return (self.pixel_to_token(video_shape[0], self.temporal_total),
                self.pixel_to_token(video_shape[1], self.spatial_total),
                self.pixel_to_token(video_shape[2], self.spatial_total))

    @staticmethod
    def pixel_to_token(num_pixel: int, pixel2token: int) -> int:
        n, rem = divmod(num_pixel, pixel2token)
        if rem > 1:
            raise ValueError(f"Number of pixel {num_pixel} is not a multiple of pixel2token {pixel2token}.")
        return n + rem  # a single leftover pixel (the first frame / causal VAE) makes one more token


hunyuan_pixel2token = Pixel2TokenFactory(temporal_vae=4, spatial_vae=8)
wan_pixel2token = Pixel2TokenFactory(temporal_vae=4, spatial_vae=8)
