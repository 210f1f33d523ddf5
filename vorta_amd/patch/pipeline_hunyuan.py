"""HunyuanVideo pipeline entry points (vorta/patch/pipeline_hunyuan.py): `sp_pipeline_call`, `vorta_pipeline_call`,
`apply_vorta_pipeline`.  Used as `pipeline.__class__.__call__ = vorta_pipeline_call`
(scripts/hunyuan/inference.py:104,120); every stock keyword is accepted unchanged, plus `self_attention_kwargs` and
`return_routing_scores`.  Returns `(video, routing_scores)` when `return_dict=False` (:470-471)."""
import os
from typing import Any, Dict, Optional

import torch

from . import _pipeline as P
from .modeling_hunyuan import apply_vorta_transformer
from .outputs import VideoPipelineOutput  # noqa: F401

try:  # remember the stock __call__ before a script replaces it
    from diffusers.pipelines.hunyuan_video.pipeline_hunyuan_video import HunyuanVideoPipeline
    P.register_pipeline_class(HunyuanVideoPipeline)
except ImportError:  # diffusers absent: register_pipeline_class(cls) must be called by the user of another class
    HunyuanVideoPipeline = None


def _decode(pipe, latents, output_type):
    """pipeline_hunyuan.py:456-459."""
    latents = latents.to(pipe.vae.dtype) / pipe.vae.config.scaling_factor
    video = pipe.vae.decode(latents, return_dict=False)[0]
    return pipe.video_processor.postprocess_video(video, output_type=output_type)


@P.mark_ours
@torch.no_grad()
def sp_pipeline_call(self, *args, self_attention_kwargs=None, **kwargs):
    """Stock denoising loop, sequence-parallel aware (pipeline_hunyuan.py:32-235).  `self_attention_kwargs` is
    accepted and ignored, for call compatibility with `vorta_pipeline_call` (:63)."""
    return P.run(self, args, kwargs, decode=_decode, self_attention_kwargs=None, return_routing_scores=False,
                 routed=False)


@P.mark_ours
@torch.no_grad()
def vorta_pipeline_call(self, *args, self_attention_kwargs: Optional[Dict[str, Any]] = None,
                        return_routing_scores: bool = False, **kwargs):
    """Denoising loop with routed sparse attention (pipeline_hunyuan.py:239-473).  The per-prompt sliding-tile
    descriptor (:378-392) is built by the transformer's forward protocol from the prompt's attention mask."""
    return P.run(self, args, kwargs, decode=_decode, self_attention_kwargs=self_attention_kwargs,
                 return_routing_scores=return_routing_scores, routed=True)


def apply_vorta_pipeline(pipeline, transformer_router_checkpoint_file: Optional[os.PathLike] = None,
                         attn_processor_kwargs: Optional[Dict[str, Any]] = None,
                         router_dtype: Optional[torch.dtype] = None):
    """pipeline_hunyuan.py:476-491 (which passes the checkpoint under a keyword its transformer patch does not
    have; here it reaches `checkpoint_file`)."""
    P.register_pipeline_class(type(pipeline))
    pipeline.__class__.__call__ = vorta_pipeline_call
    apply_vorta_transformer(pipeline.transformer, train_router=False, checkpoint_file=transformer_router_checkpoint_file,
                            attn_processor_kwargs=attn_processor_kwargs, router_dtype=router_dtype)
    return pipeline
