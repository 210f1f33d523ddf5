"""ctypes binding of libvorta_sp.so (include/vorta_sp.h): the Ulysses exchange on RCCL behind a C ABI, for callers without torch.

The Python host of this package does NOT use it -- the processors keep the reference's own interface, torch.distributed
("nccl" = RCCL), with the zero-copy receive layout of `engine.py`.  This module is what INTEGRATION.md section 2 shows a binding
to look like, and what the tests drive: `plan_seq2head` / `plan_head2seq` (host arithmetic: the list of sends and receives the
collective issues) are checked against the reference's all_to_all_4D maps for any P without a GPU; `SpComm` on a world of one
on the GPU box (RCCL refuses two ranks on one device)."""
import ctypes as C
import os
from typing import List, Tuple

_LIB = None
_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(_HERE, "csrc", "libvorta_sp.so")
UNIQUE_ID_BYTES = 128

# every symbol include/vorta_sp.h declares: (restype, argtypes)
_vp, _i32, _i64 = C.c_void_p, C.c_int32, C.c_int64


class Op(C.Structure):
    _fields_ = [("peer", _i32), ("is_send", _i32), ("offset", _i64), ("bytes", _i64)]


SYMBOLS = {
    "vorta_sp_abi_version": (C.c_int, []),
    "vorta_sp_last_error": (C.c_char_p, []),
    "vorta_sp_unique_id": (C.c_int, [_vp]),
    "vorta_sp_init": (C.c_int, [C.POINTER(_vp), _i32, _i32, _vp]),
    "vorta_sp_destroy": (C.c_int, [_vp]),
    "vorta_sp_rank": (C.c_int, [_vp]),
    "vorta_sp_size": (C.c_int, [_vp]),
    "vorta_sp_plan_seq2head": (_i64, [_i32, _i32, _i32, _i32, _i32, _i32, _i32, C.POINTER(Op), _i64]),
    "vorta_sp_plan_head2seq": (_i64, [_i32, _i32, _i32, _i32, _i32, _i32, _i32, C.POINTER(Op), _i64]),
    "vorta_sp_a2a_seq2head": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vorta_sp_a2a_head2seq": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "vorta_sp_allgather_heads": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
}


def lib():
    """the library, every declared symbol bound (a missing one raises: the header and the build must agree)"""
    global _LIB
    if _LIB is None:
        if not os.path.exists(PATH):  # not built yet: build it (hipcc + librccl; seconds), loudly if that fails
            from ..build import build_sp
            try:
                build_sp(verbose=False)
            except Exception as exc:
                raise RuntimeError(f"{PATH} is missing and could not be built (python -m vorta_amd.build; needs hipcc and librccl): {exc}")
        l = C.CDLL(PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)  # AttributeError names the missing symbol
            fn.restype, fn.argtypes = res, args
        _LIB = l
    return _LIB


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed with code {rc}: {lib().vorta_sp_last_error().decode() or 'see include/vorta_sp.h'}")


def _plan(fn, rank, P, B, H, Sl, D, elem_bytes) -> List[Tuple[int, int, int, int]]:
    n = fn(rank, P, B, H, Sl, D, elem_bytes, None, 0)
    if n < 0:
        raise ValueError(f"vorta_sp_plan: code {n} for rank {rank} of {P}, (B, H, Sl, D) = {(B, H, Sl, D)}")
    ops = (Op * max(n, 1))()
    fn(rank, P, B, H, Sl, D, elem_bytes, ops, n)
    return [(o.peer, o.is_send, o.offset, o.bytes) for o in ops[:n]]


def plan_seq2head(rank, P, B, H, Sl, D, elem_bytes=2):
    """[(peer, is_send, byte offset, bytes)] of (B,H,Sl,D) -> (B,H/P,P*Sl,D) on rank `rank` (vorta/ulysses/utils.py:15-57)"""
    return _plan(lib().vorta_sp_plan_seq2head, rank, P, B, H, Sl, D, elem_bytes)


def plan_head2seq(rank, P, B, H, Sl, D, elem_bytes=2):
    """the inverse, (B,H/P,P*Sl,D) -> (B,H,Sl,D) (utils.py:59-91)"""
    return _plan(lib().vorta_sp_plan_head2seq, rank, P, B, H, Sl, D, elem_bytes)


_DT = {"torch.bfloat16": 0, "torch.float16": 1, "torch.float32": 2, "torch.uint8": 3, "torch.float8_e4m3fn": 3, "torch.int8": 4}


class SpComm:
    """One RCCL communicator (one process per GPU).  `unique_id`: 128 bytes from `SpComm.unique_id()` on rank 0."""

    def __init__(self, rank: int, size: int, unique_id: bytes):
        if len(unique_id) != UNIQUE_ID_BYTES:
            raise ValueError("the unique id is 128 bytes")
        self._h = _vp()
        buf = C.create_string_buffer(unique_id, UNIQUE_ID_BYTES)
        check(lib().vorta_sp_init(C.byref(self._h), rank, size, C.cast(buf, _vp)), "vorta_sp_init")
        self.rank, self.size = lib().vorta_sp_rank(self._h), lib().vorta_sp_size(self._h)

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(UNIQUE_ID_BYTES)
        check(lib().vorta_sp_unique_id(C.cast(buf, _vp)), "vorta_sp_unique_id")
        return buf.raw

    def _call(self, fn, x, y, dims, what):
        import torch
        if not (x.is_cuda and y.is_cuda and x.is_contiguous() and y.is_contiguous()) or x.dtype != y.dtype:
            raise ValueError("contiguous device tensors of one dtype")
        check(fn(self._h, x.data_ptr(), y.data_ptr(), *dims, _DT[str(x.dtype)], torch.cuda.current_stream().cuda_stream), what)
        return y

    def seq2head(self, x, y=None):
        """(B,H,Sl,D) -> (B,H/P,P*Sl,D): all_to_all_4D(x, scatter_idx=1, gather_idx=2) of the reference"""
        B, H, Sl, D = x.shape
        y = x.new_empty((B, H // self.size, Sl * self.size, D)) if y is None else y
        return self._call(lib().vorta_sp_a2a_seq2head, x, y, (B, H, Sl, D), "vorta_sp_a2a_seq2head")

    def head2seq(self, x, y=None):
        """(B,H/P,S,D) -> (B,H,S/P,D): all_to_all_4D(x, scatter_idx=2, gather_idx=1)"""
        B, Hl, S, D = x.shape
        y = x.new_empty((B, Hl * self.size, S // self.size, D)) if y is None else y
        return self._call(lib().vorta_sp_a2a_head2seq, x, y, (B, Hl * self.size, S // self.size, D), "vorta_sp_a2a_head2seq")

    def allgather_heads(self, x, y=None):
        """(B,Hl,T,D) -> (B,P*Hl,T,D), rank-ordered: all_gather(x, dim=1)"""
        B, Hl, T, D = x.shape
        y = x.new_empty((B, Hl * self.size, T, D)) if y is None else y
        return self._call(lib().vorta_sp_allgather_heads, x, y, (B, Hl, T, D), "vorta_sp_allgather_heads")

    def destroy(self):
        if self._h:
            check(lib().vorta_sp_destroy(self._h), "vorta_sp_destroy")
            self._h = _vp()
