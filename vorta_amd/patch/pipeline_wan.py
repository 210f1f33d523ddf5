"""Wan pipeline entry points (vorta/patch/pipeline_wan.py): `sp_pipeline_call`, `vorta_pipeline_call`,
`apply_vorta_pipeline`; see pipeline_hunyuan.py.  Classifier-free guidance stays two batch-1 forwards per step
(pipeline_wan.py:322-344); routing scores are recorded for the conditional one only."""
import os
from typing import Any, Dict, Optional

import torch

from . import _pipeline as P
from .modeling_wan import apply_vorta_transformer
from .outputs import VideoPipelineOutput  # noqa: F401

try:
    from diffusers.pipelines.wan.pipeline_wan import WanPipeline
    P.register_pipeline_class(WanPipeline)
except ImportError:
    WanPipeline = None


def _decode(pipe, latents, output_type):
    """pipeline_wan.py:368-380: undo the latent normalisation, decode, post-process."""
    cfg = pipe.vae.config
    latents = latents.to(pipe.vae.dtype)
    mean = torch.tensor(cfg.latents_mean).view(1, cfg.z_dim, 1, 1, 1).to(latents.device, latents.dtype)
    inv_std = 1.0 / torch.tensor(cfg.latents_std).view(1, cfg.z_dim, 1, 1, 1).to(latents.device, latents.dtype)
    video = pipe.vae.decode(latents / inv_std + mean, return_dict=False)[0]
    return pipe.video_processor.postprocess_video(video, output_type=output_type)


@P.mark_ours
@torch.no_grad()
def sp_pipeline_call(self, *args, self_attention_kwargs=None, **kwargs):
    """pipeline_wan.py:25-200."""
    return P.run(self, args, kwargs, decode=_decode, self_attention_kwargs=None, return_routing_scores=False,
                 routed=False)


@P.mark_ours
@torch.no_grad()
def vorta_pipeline_call(self, *args, self_attention_kwargs: Optional[Dict[str, Any]] = None,
                        return_routing_scores: bool = False, **kwargs):
    """pipeline_wan.py:204-390."""
    return P.run(self, args, kwargs, decode=_decode, self_attention_kwargs=self_attention_kwargs,
                 return_routing_scores=return_routing_scores, routed=True)


def apply_vorta_pipeline(pipeline, transformer_router_checkpoint_file: Optional[os.PathLike] = None,
                         attn_processor_kwargs: Optional[Dict[str, Any]] = None,
                         router_dtype: Optional[torch.dtype] = None):
    """pipeline_wan.py:393-408."""
    P.register_pipeline_class(type(pipeline))
    pipeline.__class__.__call__ = vorta_pipeline_call
    apply_vorta_transformer(pipeline.transformer, train_router=False, checkpoint_file=transformer_router_checkpoint_file,
                            attn_processor_kwargs=attn_processor_kwargs, router_dtype=router_dtype)
    return pipeline
