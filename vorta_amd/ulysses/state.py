"""Sequence-parallel process-group state (mirror of vorta/ulysses/parallel_states.py:7-75)."""
import os

import torch.distributed as dist


class SequenceParallelState:
    """Process-global singleton, same attribute surface as the reference's SP_STATE:
    rank / local_rank / world_size from the torchrun environment, one SP group per `rank // sp_size`."""

    def __init__(self):
        self._reset(rank_as_group=False)

    def _reset(self, rank_as_group: bool):
        self._enabled = False
        self._sp_size = 1
        self._group_id = self.rank if rank_as_group else 0
        self._group_local_rank = 0
        self._group = None

    rank = property(lambda self: int(os.getenv("RANK", "0")))
    local_rank = property(lambda self: int(os.getenv("LOCAL_RANK", "0")))
    world_size = property(lambda self: int(os.getenv("WORLD_SIZE", "1")))
    enabled = property(lambda self: self._enabled)
    sp_size = property(lambda self: self._sp_size)
    group_id = property(lambda self: self._group_id)
    group_local_rank = property(lambda self: self._group_local_rank)
    group = property(lambda self: self._group)
    num_sp_groups = property(lambda self: self.world_size // self.sp_size)

    def setup_sp_group(self, sequence_parallel_size: int):
        if self.world_size % sequence_parallel_size != 0:
            raise ValueError(f"{self.world_size=} must be divisible by {sequence_parallel_size=}!")
        if sequence_parallel_size <= 1:
            self._reset(rank_as_group=True)
            return
        self._enabled = True
        self._sp_size = sequence_parallel_size
        self._group_id, self._group_local_rank = divmod(self.rank, sequence_parallel_size)
        # every rank must create every group (torch.distributed contract); keep our own
        for gid in range(self.num_sp_groups):
            ranks = list(range(gid * sequence_parallel_size, (gid + 1) * sequence_parallel_size))
            grp = dist.new_group(ranks)
            if gid == self._group_id:
                self._group = grp

    def cleanup(self):
        dist.destroy_process_group()
        self._reset(rank_as_group=False)


SP_STATE = SequenceParallelState()


# head -> rank placements of the zero-copy exchange (vorta_amd/attention/_sp.py says what each is)
PLACEMENTS = ("auto", "even", "uneven", "split")


def resolve_placement(name: str, H: int, P: int) -> str:
    """The placement a layer of H heads takes on P ranks: `auto` -> even when P divides H, uneven otherwise; even with
    P not dividing H is refused (there is no such exchange), anything unknown too."""
    if name not in PLACEMENTS:
        raise ValueError(f"VORTA_SP_PLACEMENT / --placement must be one of {PLACEMENTS}, got {name!r}")
    if name == "auto":
        return "even" if H % max(P, 1) == 0 else "uneven"
    if name == "even" and H % max(P, 1) != 0:
        raise ValueError(f"placement 'even' needs the rank count ({P}) to divide the heads ({H}); use 'auto' or 'uneven'")
    return name


def default_sp_groups(heads_per_rank: int, precision=False) -> int:
    """Slot groups of the exchange under `VORTA_SP_GROUPS=auto` / `bench.py --sp-groups auto` (opt-in; the default is ONE group
    until a node run has measured the links): the heads of a rank travel in that many groups so that the exchange
    of one overlaps the attention of another.  Chosen from the one-GPU
    emulation of the heaviest rank of 8 with an ASSUMED wire (60 / 150 GB/s per xGMI link + 10 us per collective;
    profiles/r05_sp_groups_emulated.txt) -- the RCCL transport has not run on a node yet:
      3 heads per rank (Hunyuan, 24 / 8): 1 group (2 / 3 groups lose 2-3 % at 150 GB/s, gain 2 % at 60);
      5 heads per rank, 16-bit / fp8pv (Wan-14B, 40 / 8): 2 groups (-4 % at 150 GB/s, -9 % at 60);
      5 heads per rank, int8 / e4m3 scores (the attention is 1.6-1.8 x shorter, the wire is not): 3 groups (-2 % / -9 %)."""
    if heads_per_rank < 4:
        return 1
    if precision in ("i8pv", "auto8", True, "fp8") and heads_per_rank >= 5:  # ("auto8" runs the int8-score kernels)
        return 3
    return 2
