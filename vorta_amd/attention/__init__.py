"""Operator surface of vorta.attention (vorta/attention/__init__.py:1-16), MI355X-native."""
from .coreset_select import (LowresGroupInfo, get_group_info, pool_sequence_by_similarity,
                             unpool_sequence_by_similarity)
from .hunyuan import (HunyuanVideoFlashAttnProcessor, HunyuanVideoFlashAttnProcessorTripleEval,
                      HunyuanVideoFlashAttnProcessorTripleTrain)
from .sliding_tile import SlidingTileDescriptor, create_sliding_tile_attn_mask_func
from .wan import WanAttnProcessor2_0, WanAttnProcessorTripleEval, WanAttnProcessorTripleTrain

__all__ = ["LowresGroupInfo", "get_group_info", "pool_sequence_by_similarity", "unpool_sequence_by_similarity",
           "HunyuanVideoFlashAttnProcessor", "HunyuanVideoFlashAttnProcessorTripleEval",
           "HunyuanVideoFlashAttnProcessorTripleTrain", "create_sliding_tile_attn_mask_func", "SlidingTileDescriptor",
           "WanAttnProcessor2_0", "WanAttnProcessorTripleEval", "WanAttnProcessorTripleTrain"]
