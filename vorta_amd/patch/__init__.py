"""vorta.patch: Router, keyword preparation, pixel->token map, and the model / pipeline entry points.

The scripts import the entry points from the submodules, as with the reference (whose `__init__` is empty):
`vorta.patch.modeling_{hunyuan,wan}.{apply_vorta_transformer, apply_sp_flashattn_transformer}`,
`vorta.patch.pipeline_{hunyuan,wan}.{vorta_pipeline_call, sp_pipeline_call, apply_vorta_pipeline}`
(scripts/hunyuan/inference.py:27-32, scripts/wan/inference.py:32-37).  They attach to the stock diffusers
forwards with hooks (vorta_amd/patch/_engine.py) and import nothing from diffusers at module level."""
from .router import Router, load_router_checkpoint
from .utils import (Pixel2TokenFactory, hunyuan_pixel2token, prepare_hunyuan_self_attn_kwargs,
                    prepare_wan_self_attn_kwargs, wan_pixel2token)
