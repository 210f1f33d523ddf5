"""vorta.patch surface needed by the hot path: Router, kwargs preparation, pixel->token map.

The model / pipeline monkey-patches (vorta/patch/modeling_*.py, pipeline_*.py) are callers of the hot path
that need the `diffusers` classes; they are the next row of SURVEY.md §8(f) (N3), not part of this round."""
from .router import Router, load_router_checkpoint
from .utils import (Pixel2TokenFactory, hunyuan_pixel2token, prepare_hunyuan_self_attn_kwargs,
                    prepare_wan_self_attn_kwargs, wan_pixel2token)
