"""Ulysses sequence parallelism (mirror of vorta/ulysses/__init__.py:10-21, inference subset + the
zero-copy engine).  Training-only helpers of the reference (`all_to_all` on lists, `reduce_loss`,
`TrainingLog`, `set_seed`) are out of scope."""
from .comm import all_gather, all_to_all_4D, broadcast_sp_group, dist_prefix, shrink_dim
from .engine import UlyssesLayout, UlyssesRoutedAttention, balanced_head_order, make_row_map
from .state import SP_STATE, SequenceParallelState

__all__ = ["SP_STATE", "SequenceParallelState", "all_to_all_4D", "all_gather", "shrink_dim", "broadcast_sp_group",
           "dist_prefix", "UlyssesLayout", "UlyssesRoutedAttention", "balanced_head_order", "make_row_map"]
