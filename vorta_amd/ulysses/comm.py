"""Ulysses collectives with the reference's call surface (vorta/ulysses/utils.py), forward only.

`torch.distributed` is the transport: backend "nccl" is RCCL over xGMI on the GPU box, "gloo" on CPU for the
world_size>1 tests.  Differences from the reference, by design:
  * no `torch.cuda.synchronize()` after the collective (utils.py:49,81) -- stream order is enough;
  * one packing copy per call instead of two transpose+contiguous passes per side (for batch 1 the
    seq->head direction sends the input buffer as is, the head->seq direction receives in place);
  * the attention processors do not use these at all when SP is on: they use the zero-copy path of
    engine.py.  These functions exist so code written against the reference keeps working.
Backward passes (SeqAllToAll4D.backward, AllGather.backward) are training-only and out of scope.
"""
import torch
import torch.distributed as dist

from .state import SP_STATE


def _world(group) -> int:
    return dist.get_world_size(group) if dist.is_initialized() else 1


def _all_to_all_4D(x: torch.Tensor, scatter_idx: int = 2, gather_idx: int = 1, group=None) -> torch.Tensor:
    """(scatter 1, gather 2): (B, H, S/P, D) -> (B, H/P, S, D); (scatter 2, gather 1): the inverse.
    Rank r ends up with the contiguous head block [r*H/P, (r+1)*H/P) and the rank-major concatenation of
    the sequence shards (vorta/ulysses/utils.py:15-93; verified by tests/golden/g9)."""
    assert x.dim() == 4, f"input must be 4D tensor, got {x.dim()} and shape {x.shape}"
    P = _world(group)
    if (scatter_idx, gather_idx) == (1, 2):
        B, H, Sl, D = x.shape
        Hl = H // P
        # chunk j (for rank j) = heads [j*Hl, (j+1)*Hl): contiguous for B == 1, one packing copy otherwise
        send = x.reshape(B, P, Hl, Sl, D).transpose(0, 1).contiguous()  # (P, B, Hl, Sl, D); a view copy-free if B == 1
        recv = torch.empty_like(send)
        if P > 1:
            dist.all_to_all_single(recv, send, group=group)
        else:
            recv = send
        # (P_src, B, Hl, Sl, D) -> (B, Hl, P_src*Sl, D)
        return recv.permute(1, 2, 0, 3, 4).reshape(B, Hl, P * Sl, D)
    if (scatter_idx, gather_idx) == (2, 1):
        B, Hl, S, D = x.shape
        Sl = S // P
        send = x.reshape(B, Hl, P, Sl, D).permute(2, 0, 1, 3, 4).contiguous()  # (P, B, Hl, Sl, D)
        recv = torch.empty_like(send)
        if P > 1:
            dist.all_to_all_single(recv, send, group=group)
        else:
            recv = send
        # (P_src, B, Hl, Sl, D) -> (B, P_src*Hl, Sl, D): free for B == 1
        return recv.transpose(0, 1).reshape(B, P * Hl, Sl, D)
    raise RuntimeError("scatter_idx must be 1 or 2 and gather_idx must be 1 or 2")


def all_to_all_4D(input_: torch.Tensor, scatter_idx: int, gather_idx: int) -> torch.Tensor:
    return _all_to_all_4D(input_, scatter_idx, gather_idx, group=SP_STATE.group)


def all_gather(input_: torch.Tensor, dim: int = 0) -> torch.Tensor:
    """rank-ordered concatenation along `dim` (vorta/ulysses/utils.py:127-162)."""
    if not SP_STATE.enabled:
        return input_
    parts = [torch.empty_like(input_) for _ in range(SP_STATE.sp_size)]
    dist.all_gather(parts, input_.contiguous(), group=SP_STATE.group)
    return torch.cat(parts, dim=dim)


def shrink_dim(tensor: torch.Tensor, dim: int) -> torch.Tensor:
    """this rank's 1/P slice along `dim` (vorta/ulysses/utils.py:218-223)."""
    if not SP_STATE.enabled:
        return tensor
    n = tensor.size(dim) // SP_STATE.sp_size
    return tensor.narrow(dim, n * SP_STATE.group_local_rank, n)


def broadcast_sp_group(input_: torch.Tensor):
    """broadcast from the first rank of the SP group (vorta/ulysses/utils.py:225-227)."""
    dist.broadcast(input_, src=SP_STATE.group_id * SP_STATE.sp_size, group=SP_STATE.group)


def dist_prefix(msg: str) -> str:
    s = SP_STATE
    return (f"[Rank: {s.local_rank}/{s.rank}/{s.world_size} | DP: {s.group_id}/{s.num_sp_groups} | "
            f"SP: {s.group_local_rank}/{s.sp_size}] {msg}")


def set_seed(seed: int, device_specific: bool = False) -> None:
    """Seed `random`, `numpy` and `torch` (vorta/ulysses/utils.py:238-257); `device_specific` offsets by the global
    rank."""
    import random

    import numpy as np
    if device_specific:
        seed += SP_STATE.rank
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
