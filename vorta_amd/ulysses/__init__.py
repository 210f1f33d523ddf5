"""Ulysses sequence parallelism (mirror of vorta/ulysses/__init__.py:10-21, inference subset + the
zero-copy engine).  Training-only helpers of the reference (the autograd `all_to_all`, `reduce_loss`,
`TrainingLog`) are out of scope."""
from .comm import all_gather, all_to_all_4D, broadcast_sp_group, dist_prefix, set_seed, shrink_dim
from .engine import (UlyssesLayout, UlyssesRoutedAttention, balanced_head_order, balanced_placement, exchange_and_attend,
                     exchange_selfcheck, make_row_map, placement_loads, slot_groups, split_placement, tag_rows)
from .state import SP_STATE, SequenceParallelState

__all__ = ["SP_STATE", "SequenceParallelState", "all_to_all_4D", "all_gather", "shrink_dim", "broadcast_sp_group",
           "dist_prefix", "set_seed", "UlyssesLayout", "UlyssesRoutedAttention", "balanced_head_order", "balanced_placement", "make_row_map", "exchange_and_attend", "slot_groups",
           "exchange_selfcheck", "tag_rows", "split_placement", "placement_loads"]
