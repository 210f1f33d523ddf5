"""Tensor-level launchers over the C ABI (vorta_amd/_C.py).  PyTorch supplies device memory and the current
stream; all arithmetic happens inside libvorta_hip.so.

Tensors are (H, S, D) views ("head-major"): a (B, H, S, D) batch is folded with `fold_heads`.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Sequence, Tuple

import torch

from . import _C

_DT = {torch.bfloat16: _C.VORTA_BF16, torch.float16: _C.VORTA_FP16}
FP8_STORAGE = torch.uint8  # e4m3 bytes travel as uint8 tensors (no arithmetic is ever done on them in torch)


class Timeline:
    """Optional per-launch HIP-event timing of vorta_attn_fwd (used by bench.py for the roofline figure).
    Events are recorded on the stream the kernels are launched on (torch's current stream)."""

    def __init__(self):
        self.records = []  # (tag, kernel symbol, n_workgroups, flops, start_event, end_event)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for tag, br, nwg, flops, e0, e1 in self.records:
            d = out.setdefault((tag, br), dict(launches=0, ms=0.0, flops=0.0, workgroups=0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += flops
            d["workgroups"] += nwg
        return out


_timeline: Optional[Timeline] = None
from ._debug import flag as _debug_flag  # (A/B switches: one gate, VORTA_DEBUG)
DEFAULT_VARIANT = int(_debug_flag("attn_variant", "0"))
NO_XCD_REMAP = int(_debug_flag("no_xcd_remap", "0"))


def set_timeline(t: Optional[Timeline]):
    global _timeline
    _timeline = t


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _require_gpu(*ts: torch.Tensor):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _C.VortaHipError("vorta_amd ops need tensors on an MI355X (cuda) device; there is no CPU path")


def _tensor(t: torch.Tensor) -> _C.Tensor:
    if t.dim() != 3 or t.stride(2) != 1:
        raise ValueError(f"expected a (H,S,D) view with contiguous D, got shape {tuple(t.shape)} strides {t.stride()}")
    return _C.Tensor(t.data_ptr(), t.stride(0), t.stride(1))


def _ptr(t: Optional[torch.Tensor], dtype=torch.int32):
    if t is None:
        return None
    if t.dtype != dtype or not t.is_contiguous():
        raise ValueError(f"index tables must be contiguous {dtype}")
    return t.data_ptr()


def fold_heads(x: torch.Tensor) -> torch.Tensor:
    """(B,H,S,D) -> (B*H,S,D) view without copying (requires stride_b == H*stride_h)."""
    if x.dim() == 3:
        return x
    B, H, S, D = x.shape
    if B == 1:
        return x[0]
    if x.stride(0) != H * x.stride(1):
        raise ValueError("batch cannot be folded into the head axis without a copy")
    return x.as_strided((B * H, S, D), (x.stride(1), x.stride(2), x.stride(3)), x.storage_offset())


def _attn_args(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, *,
               n_q: int, n_kv: int,
               head_list: Optional[torch.Tensor] = None, n_heads: Optional[int] = None,
               n_heads_dev: Optional[torch.Tensor] = None,
               q_group_len: int = 0, q_row_offset: int = 0, q_valid: Optional[int] = None,
               q_rows: Optional[torch.Tensor] = None,
               kv_row_offset: int = 0, kv_rows: Optional[torch.Tensor] = None, kv_rows_stride_g: int = 0,
               dup_rows: Optional[torch.Tensor] = None, n_dup_pos: int = 0,
               scale: Optional[float] = None, block_rows: int = 0, n_splits: int = 1,
               n_kv_dev: Optional[torch.Tensor] = None, q_valid_dev: Optional[torch.Tensor] = None,
               variant: int = 0, tag: str = "", flops: float = 0.0,
               v_descale: Optional[torch.Tensor] = None, fp8_opts: Optional[dict] = None,
               q_block_table: Optional[torch.Tensor] = None, n_key_lists: int = 0,
               i8: Optional["I8Operands"] = None):
    """Fill a vorta_attn_args; returns (args, workspace tensors to keep alive until the launch is enqueued).
    q,k,v of dtype uint8 = e4m3 operands from `fp8_quantize_qkv` (then `v_descale` is required and `out` is 16-bit):
    the args carry `_ext`, the vorta_attn_fp8_ext of the fp8 entry points.
    k of dtype int8 = the rows of `i8_quantize_k` (then `i8` = its I8Operands, or views of them with k's geometry, is
    required, q and out are 16-bit, v is e4m3): `_ext` is the vorta_attn_i8_ext of the int8-score entry points."""
    _require_gpu(q, k, v, out)
    fp8 = q.dtype == FP8_STORAGE
    i8ops = i8
    i8 = (not fp8) and k.dtype == torch.int8  # int8 scores, e4m3 P V (csrc/attn_fwd_i8.hip)
    mixed = (not fp8) and (not i8) and v.dtype == FP8_STORAGE  # 16-bit scores, e4m3 P V (csrc/attn_fwd_mx.hip)
    if i8:
        if q.dtype not in _DT or q.dtype != out.dtype or v.dtype != FP8_STORAGE:
            raise ValueError("int8-score attention takes 16-bit q and out of one dtype, int8 k and e4m3 (uint8) v")
        kb, qp, hs = (None, None, None) if i8ops is None else (i8ops.k_bias, i8ops.q_prep, i8ops.k_head_scale)
        if kb is None or kb.dtype != torch.float32 or kb.dim() != 2 or kb.stride(1) != 1 or kb.shape[0] < k.shape[0] \
                or kb.shape[1] < k.shape[1]:
            raise ValueError("int8-score attention needs i8.k_bias: float32 (heads, rows) with unit row stride (i8_quantize_k)")
        if qp is None or qp.dtype != torch.float32 or qp.dim() != 3 or tuple(qp.shape[1:]) != (2, q.shape[-1]) \
                or not qp.is_contiguous() or hs is None or hs.dtype != torch.float32 or not hs.is_contiguous():
            raise ValueError("int8-score attention needs i8.q_prep (heads, 2, D) and i8.k_head_scale (heads,) float32 from i8_quantize_k")
    if fp8:
        if not (k.dtype == v.dtype == FP8_STORAGE) or out.dtype not in _DT:
            raise ValueError("fp8 attention takes e4m3 (uint8) q,k,v and a bf16 / fp16 output")
    elif mixed:
        if q.dtype not in _DT or not (q.dtype == k.dtype == out.dtype):
            raise ValueError("mixed-precision attention takes 16-bit q, k, out of one dtype and an e4m3 (uint8) v")
    elif i8:
        pass  # checked above
    elif q.dtype not in _DT or not (q.dtype == k.dtype == v.dtype == out.dtype):
        raise ValueError("q,k,v,out must share dtype bf16 or fp16")
    if fp8 or mixed or i8:
        if v_descale is None or v_descale.dtype != torch.float32 or v_descale.dim() != 2 or v_descale.shape[1] != q.shape[-1] \
                or not v_descale.is_contiguous():
            raise ValueError("fp8 attention needs v_descale: contiguous float32 (heads, D) from fp8_quantize_qkv / fp8_quantize_v")
    a = _C.AttnArgs()
    a.struct_size = C.sizeof(_C.AttnArgs)
    a.dtype = _C.VORTA_FP8E4M3 if fp8 else _DT[q.dtype]
    a._ext = None
    a._mixed = mixed
    a._i8 = i8
    if i8:
        ext = _C.AttnI8Ext()
        ext.struct_size = C.sizeof(_C.AttnI8Ext)
        ext.k_bias, ext.k_bias_stride_h = i8ops.k_bias.data_ptr(), i8ops.k_bias.stride(0)
        ext.q_prep, ext.q_prep_stride_h = i8ops.q_prep.data_ptr(), i8ops.q_prep.stride(0)
        ext.k_head_scale = i8ops.k_head_scale.data_ptr()
        a._i8ops = i8ops  # (keeps the operand tensors alive until the launch is enqueued)
        ext.v_descale, ext.v_descale_stride_h = v_descale.data_ptr(), v_descale.stride(0)
        o = fp8_opts or FP8_OPTS
        ext.p_bias, ext.defer = float(o.get("p_bias", 0.0)), float(o.get("defer", 0.0))
        a._ext = ext
    elif fp8 or mixed:
        ext = _C.AttnFp8Ext()
        ext.struct_size = C.sizeof(_C.AttnFp8Ext)
        ext.out_dtype = _DT[out.dtype]
        ext.v_descale, ext.v_descale_stride_h = v_descale.data_ptr(), v_descale.stride(0)
        o = fp8_opts or FP8_OPTS
        ext.p_bias, ext.defer, ext.flags = float(o.get("p_bias", 0.0)), float(o.get("defer", 0.0)), int(o.get("flags", 0))
        if mixed:
            ext.flags |= 2
        a._ext = ext
    a.head_dim = q.shape[-1]
    a.q, a.k, a.v, a.o = _tensor(q), _tensor(k), _tensor(v), _tensor(out)
    if n_heads is None:
        n_heads = head_list.numel() if head_list is not None else q.shape[0]
    a.n_heads = n_heads
    a.head_list = _ptr(head_list)
    a.n_heads_dev = _ptr(n_heads_dev)
    a.n_q, a.q_group_len, a.q_row_offset = n_q, q_group_len, q_row_offset
    a.q_valid = n_q if q_valid is None else q_valid
    # the kernels trust the tables: check every extent that can be checked on the host (the row VALUES live on the
    # device and are the caller's contract, vorta_hip.h)
    n_groups = 1 if q_group_len <= 0 else -(-n_q // q_group_len)
    if q_block_table is not None:
        # query groups of different lengths: one (group, first, end) row per workgroup (vorta_hip.h); n_key_lists = the
        # number of groups the rows refer to (extent check of kv_rows below)
        if q_block_table.dim() != 2 or q_block_table.shape[1] != 3 or block_rows not in (128, 256) or n_key_lists <= 0:
            raise ValueError("q_block_table must be (n_blocks, 3) int32 with an explicit block_rows (128 / 256) and n_key_lists")
        n_groups = n_key_lists
    if head_list is not None and head_list.numel() < n_heads:
        raise ValueError(f"head_list holds {head_list.numel()} heads, n_heads = {n_heads}")
    if q_rows is not None and (q_rows.dim() not in (1, 2) or q_rows.shape[-1] < n_q
                               or (q_rows.dim() == 2 and q_rows.shape[0] < n_heads)):
        raise ValueError(f"q_rows {tuple(q_rows.shape)} does not cover n_q = {n_q} rows for {n_heads} head slots")
    if kv_rows is not None:
        if kv_rows_stride_g > 0:
            need = (n_groups - 1) * kv_rows_stride_g + n_kv
            if kv_rows.numel() < need or kv_rows_stride_g < 0:
                raise ValueError(f"kv_rows {tuple(kv_rows.shape)} does not cover {n_groups} groups x {n_kv} keys "
                                 f"at group stride {kv_rows_stride_g}")
        elif n_groups > 1:
            raise ValueError("query groups need kv_rows_stride_g (one key list per group)")
        elif kv_rows.dim() not in (1, 2) or kv_rows.shape[-1] < n_kv or (kv_rows.dim() == 2 and kv_rows.shape[0] < n_heads):
            raise ValueError(f"kv_rows {tuple(kv_rows.shape)} does not cover n_kv = {n_kv} keys for {n_heads} head slots "
                             f"(1-D: shared by all heads; 2-D: one row per head slot)")
    if dup_rows is not None and (dup_rows.dim() not in (2, 3) or (dup_rows.dim() == 3 and dup_rows.shape[0] < n_heads)
                                 or dup_rows.shape[-2] < (n_dup_pos or dup_rows.shape[-2])):
        raise ValueError(f"dup_rows {tuple(dup_rows.shape)} does not cover {n_dup_pos} positions for {n_heads} head slots")
    a.q_rows = _ptr(q_rows)
    a.q_rows_stride_h = q_rows.stride(0) if (q_rows is not None and q_rows.dim() == 2) else 0
    a.n_kv, a.kv_row_offset = n_kv, kv_row_offset
    a.kv_rows = _ptr(kv_rows)
    if kv_rows is not None and kv_rows_stride_g == 0 and kv_rows.dim() == 2:
        a.kv_rows_stride_h = kv_rows.stride(0)  # (slots, n_kv): per head slot, one group
    a.kv_rows_stride_g = kv_rows_stride_g
    a.dup_rows = _ptr(dup_rows)
    if dup_rows is not None:
        if dup_rows.dim() == 3:  # (slots, n_dup_pos, n_dup)
            a.dup_rows_stride_h = dup_rows.stride(0)
        a.n_dup = dup_rows.shape[-1]
        a.n_dup_pos = n_dup_pos if n_dup_pos else dup_rows.shape[-2]
    a.scale = (1.0 / math.sqrt(q.shape[-1])) if scale is None else scale
    a.block_rows = block_rows
    a.n_splits = n_splits
    a.n_kv_dev, a.q_valid_dev = _ptr(n_kv_dev), _ptr(q_valid_dev)
    a.variant = 0 if (fp8 or mixed or i8) else (variant or DEFAULT_VARIANT)
    if a.variant != 1:
        # the pipelined kernel's 32-bit K/V offsets (vorta_hip.h): beyond a 2 GiB window per head, or 2^24 rows, use
        # the plain HIP kernel (64-bit addressing) instead
        for t in (k, v):
            esz = t.element_size()
            if t.shape[1] >= (1 << 24) or t.shape[1] * t.stride(1) * esz > 0x7fffffff or t.stride(1) * esz >= (1 << 24):
                if fp8 or mixed or i8:
                    raise ValueError("fp8 attention addresses K/V rows with 32-bit offsets: a head must fit a 2 GiB window")
                a.variant = 1
    a.reserved = NO_XCD_REMAP
    if q_block_table is not None:
        a.q_block_table, a.n_q_blocks = _ptr(q_block_table), q_block_table.shape[0]
    ws = None
    if n_splits > 1:
        so, sm = C.c_uint64(), C.c_uint64()
        _C.check(_C.lib().vorta_attn_workspace_bytes(C.byref(a), C.byref(so), C.byref(sm)), "vorta_attn_workspace_bytes")
        ws = (torch.empty(so.value // 4, dtype=torch.float32, device=q.device),
              torch.empty(sm.value // 4, dtype=torch.float32, device=q.device))
        a.ws_o, a.ws_ml = ws[0].data_ptr(), ws[1].data_ptr()
    return a, ws


def _launch_one(a):
    if a._i8:
        _C.check(_C.lib().vorta_attn_fwd_i8(C.byref(a), C.byref(a._ext), _stream()), "vorta_attn_fwd_i8")
    elif a._ext is not None:
        _C.check(_C.lib().vorta_attn_fwd_fp8(C.byref(a), C.byref(a._ext), _stream()), "vorta_attn_fwd_fp8")
    else:
        _C.check(_C.lib().vorta_attn_fwd(C.byref(a), _stream()), "vorta_attn_fwd")


def _plan(a) -> Tuple[int, int, str]:
    br, nwg, kid = C.c_int32(), C.c_int64(), C.c_int32()
    _C.check(_C.lib().vorta_attn_plan(C.byref(a), C.byref(br), C.byref(nwg), C.byref(kid)), "vorta_attn_plan")
    tname = "_Float16" if a.dtype == _C.VORTA_FP16 else "__bf16"
    nw, kk = kid.value // 16, kid.value % 16
    if a._i8:
        return br.value, nwg.value, f"attn_i8_kernel<{tname},{nw},{'true' if kk & 2 else 'false'}>"
    if a._ext is not None:
        tname = "_Float16" if a._ext.out_dtype == _C.VORTA_FP16 else "__bf16"
        if a._ext.flags & 2:
            return br.value, nwg.value, f"attn_mx_kernel<{tname},{nw},{'true' if kk & 2 else 'false'}>"
        return br.value, nwg.value, (f"attn8_kernel<{tname},{nw},{'true' if kk & 2 else 'false'},"
                                     f"{'false' if a._ext.flags & 1 else 'true'}>")
    sym = (f"attn_fwd_pipe_kernel<{tname},{nw},{'true' if kk & 2 else 'false'}>" if kk & 1
           else f"attn_fwd_kernel<{tname},{nw}>")
    return br.value, nwg.value, sym


def attn_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, *, tag: str = "",
             flops: float = 0.0, **kw) -> None:
    """vorta_attn_fwd (include/vorta_hip.h); keyword arguments as `_attn_args`.  q_rows/kv_rows/dup_rows: int32; a
    leading head-slot axis is optional (2-D q_rows = per head slot, 1-D = shared)."""
    a, ws = _attn_args(q, k, v, out, **kw)
    if _timeline is not None:
        _, nwg, sym = _plan(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _launch_one(a)
        e1.record()
        _timeline.records.append((tag, sym, nwg, flops, e0, e1))
        return
    _launch_one(a)
    # `ws` may be released now: the caching allocator is stream ordered and the launch is on this stream


MAX_FUSED = 6  # include/vorta_hip.h VORTA_MAX_FUSED_LAUNCHES (the library refuses more with VORTA_EINVAL)


def attn_fwd_batch(calls) -> None:
    """vorta_attn_fwd_batch: `calls` is a list of dicts of attn_fwd arguments (q,k,v,out + keywords, optional tag /
    flops).  Launches that all resolve to the 256-row pipelined kernel are fused into one grid, in list order
    (put the longest key loops first); otherwise they are launched one by one.
    Two optional per-call hints apply ONLY to a call that does enter a fused grid (a stand-alone launch keeps the
    arguments it was given): `fused_n_splits` = at most this many key splits there (inside a fused grid a few-row launch
    need not fill the chip by itself), `fused_first` = dispatch it ahead of the other launches of the grid."""
    built, hints = [], []

    def build(c, n_splits=None):
        kw = {key: val for key, val in c.items() if key not in ("q", "k", "v", "out")}
        if n_splits is not None:
            kw["n_splits"] = n_splits
        return _attn_args(c["q"], c["k"], c["v"], c["out"], **kw)

    for c in calls:
        c = dict(c)
        tag, flops = c.pop("tag", ""), c.pop("flops", 0.0)
        splits, first = c.pop("fused_n_splits", None), bool(c.pop("fused_first", False))
        own = c.get("n_splits", 1)
        hinted = splits is not None and own > max(1, splits)
        hints.append((hinted, first, c))
        # built for the fused grid first (the usual outcome; the stand-alone form would allocate a split-key workspace of
        # hundreds of MB just to drop it)
        a, ws = build(c, max(1, splits) if hinted else None)
        built.append((a, ws, tag, flops))
    if not built:
        return

    def fusable_one(a):
        return _plan(a)[0] == 256 and a.variant != 1 and not (a._ext is not None and not a._i8 and a._ext.flags & 1)

    ok = [fusable_one(a) for a, _, _, _ in built]
    fuse_all = 1 < len(built) <= MAX_FUSED and all(ok)
    fuse_some = not fuse_all and 2 <= sum(ok) <= MAX_FUSED and sum(ok) < len(built)
    in_grid = [o and (fuse_all or fuse_some) for o in ok]
    for i, (hinted, _, c) in enumerate(hints):
        if hinted and not in_grid[i]:  # stand-alone after all: the caller's own split count
            a, ws = build(c)
            built[i] = (a, ws, built[i][2], built[i][3])
    if fuse_all or fuse_some:
        fused = sorted((i for i in range(len(built)) if in_grid[i]), key=lambda i: (not hints[i][1], i))
        attn_fwd_batch_built([built[i] for i in fused])
        # mixed workgroup sizes: the others run on their own, in list order after the fused grid
        attn_fwd_batch_built([b for b, g in zip(built, in_grid) if not g], fuse=False)
        return
    attn_fwd_batch_built(built, fuse=False)


def attn_fwd_batch_built(built, fuse: bool = True) -> None:
    fusable = fuse and len(built) > 1
    if not fusable:
        for a, ws, tag, flops in built:
            if _timeline is not None:
                _, nwg, sym = _plan(a)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _launch_one(a)
                e1.record()
                _timeline.records.append((tag, sym, nwg, flops, e0, e1))
            else:
                _launch_one(a)
        return
    arr = (_C.AttnArgs * len(built))(*[a for a, _, _, _ in built])
    ext = built[0][0]._ext  # one operand set per layer: every fused launch shares v_descale and the options
    i8 = built[0][0]._i8
    if any(a._i8 != i8 for a, _, _, _ in built) or (not i8 and any(
            (a._ext is None) != (ext is None) or (ext is not None and (a._ext.flags & 2) != (ext.flags & 2)) for a, _, _, _ in built)):
        raise ValueError("a fused grid is all 16-bit, all e4m3, all mixed-precision or all int8-score")

    def go():
        if i8:
            _C.check(_C.lib().vorta_attn_fwd_batch_i8(arr, C.byref(ext), len(built), _stream()), "vorta_attn_fwd_batch_i8")
        elif ext is not None:
            _C.check(_C.lib().vorta_attn_fwd_batch_fp8(arr, C.byref(ext), len(built), _stream()), "vorta_attn_fwd_batch_fp8")
        else:
            _C.check(_C.lib().vorta_attn_fwd_batch(arr, len(built), _stream()), "vorta_attn_fwd_batch")

    if _timeline is not None:
        if i8:
            sym = f"attn_i8_multi_kernel<{'_Float16' if built[0][0].dtype == _C.VORTA_FP16 else '__bf16'}>"
        elif ext is not None:
            sym = ("attn_mx_multi_kernel" if ext.flags & 2 else "attn8_multi_kernel") + \
                f"<{'_Float16' if ext.out_dtype == _C.VORTA_FP16 else '__bf16'}>"
        else:
            sym = f"attn_fwd_multi_kernel<{'_Float16' if built[0][0].dtype == _C.VORTA_FP16 else '__bf16'}>"
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        go()
        e1.record()
        _timeline.records.append(("+".join(t for _, _, t, _ in built), sym,
                                  sum(_plan(a)[1] for a, _, _, _ in built), sum(f for _, _, _, f in built), e0, e1))
        return
    go()


FP8_OPTS: dict = {}  # process-wide defaults of the fp8 kernels' options (p_bias, defer, flags); experiments only


class Fp8Operands:
    """e4m3 copies of one layer's q,k,v (uint8 storage, same (H,S,D) geometry) + the per-channel v descale."""
    __slots__ = ("q", "k", "v", "v_descale", "ws")

    def __init__(self, q, k, v, v_descale, ws):
        self.q, self.k, self.v, self.v_descale, self.ws = q, k, v, v_descale, ws

    def multipliers(self):
        """(qmul (H,), kmul (H,), vmul (H,D)) as written by the scales kernel (device tensors; tests / diagnostics)."""
        H, D = self.v_descale.shape
        base = 2 * H + H * D
        return self.ws[base:base + H], self.ws[base + H:base + 2 * H], self.ws[base + 2 * H:base + 2 * H + H * D].view(H, D)

    def k_center(self):
        """(H,D) vector subtracted from every key of a head before the conversion (zeros unless centring was asked for)"""
        H, D = self.v_descale.shape
        base = 2 * (2 * H + H * D)
        return self.ws[base:base + H * D].view(H, D)


def fp8_ws_partials(ws: torch.Tensor, heads: int, head_dim: int = 128) -> torch.Tensor:
    """the sample-partials region of a quantiser workspace (vorta_fp8_quant_ws_partials): what the ranks zero, fill with
    `fp8_quantize_qkv(phase="stats")` and all-reduce with SUM"""
    first, count = C.c_int64(), C.c_int64()
    _C.check(_C.lib().vorta_fp8_quant_ws_partials(heads, head_dim, C.byref(first), C.byref(count)), "vorta_fp8_quant_ws_partials")
    return ws[first.value:first.value + count.value]


def fp8_quantize_qkv(q: torch.Tensor, k: torch.Tensor, v: Optional[torch.Tensor], scale: Optional[float] = None, *,
                     out: Optional[Fp8Operands] = None, v_per_head: bool = False, center_k: bool = False,
                     heads: Optional[int] = None, seg_len: int = 0, tail_first: int = 0, tail_len: int = 0,
                     slots: Optional[Tuple[int, int]] = None, video_tokens: int = 0, phase: Optional[str] = None,
                     token_offset: int = 0, total_tokens: int = 0, src_map: Optional[torch.Tensor] = None) -> Fp8Operands:
    """vorta_fp8_quantize_qkv: (H,S,D) bf16/fp16 views -> e4m3 copies (contiguous (H,S,D) uint8) with the softmax scale
    and log2(e) folded into q/k.  `out` = a previous result to overwrite (same shapes).  `center_k`: subtract a per-head
    centre from the keys first (softmax-invariant; see include/vorta_hip.h).  `seg_len > 0`: q,k,v are (1,rows,D) row
    arrays in which row r belongs to head (r // seg_len) % heads (the Ulysses receive layout); from row `tail_first` on
    only the first `tail_len` rows of a segment hold data; `slots` = (first, end): only those head slots of it.
    `v=None`: q and k only (flags bit2) -- `out.v` / `out.v_descale` are left to `fp8_v_convert`.
    `video_tokens`: tokens of a head before its tail (text) tokens (0: all) -- the sample is summed in 8 ranges of them + the
    tail, pass the same value wherever the same heads are converted.  Sequence shards (q, k only): `phase="stats"` adds the
    sample partials of tokens [token_offset, token_offset + S) of `total_tokens` to `out.ws` (zero `fp8_ws_partials(out)`
    first, all-reduce it with SUM afterwards), `phase="convert"` converts them with the scales of the whole sequence,
    output head h <- head src_map[h]: the two phases write the bytes one plain call over the assembled sequence writes."""
    skip_v = v is None
    if skip_v:
        if out is None:
            raise ValueError("fp8_quantize_qkv(v=None) needs `out`: its v / v_descale are filled by fp8_v_convert")
        v = q
    _require_gpu(q, k, v)
    if q.dtype not in _DT or not (q.dtype == k.dtype == v.dtype):
        raise ValueError("q,k,v must share dtype bf16 or fp16")
    if not (q.shape == k.shape == v.shape) or q.dim() != 3:
        raise ValueError("fp8_quantize_qkv takes (H,S,D) views of equal shape")
    Hx, S, D = q.shape
    if seg_len > 0:
        if Hx != 1 or not heads or heads < 1:
            raise ValueError("fp8_quantize_qkv: the segmented layout takes (1,rows,D) row arrays and the number of heads")
        H = heads
    else:
        if heads not in (None, Hx):
            raise ValueError("fp8_quantize_qkv: `heads` only applies to the segmented layout")
        H = Hx
    dev = q.device
    nws = _C.lib().vorta_fp8_quant_ws_floats(H, D)
    if nws < 0:
        raise ValueError(f"fp8_quantize_qkv: unsupported head_dim {D}")
    if out is None:
        out = Fp8Operands(*(torch.empty((Hx, S, D), dtype=FP8_STORAGE, device=dev) for _ in range(3)),
                          torch.empty((H, D), dtype=torch.float32, device=dev),
                          torch.zeros(nws, dtype=torch.float32, device=dev))
    a = _C.Fp8QuantArgs()
    a.struct_size = C.sizeof(_C.Fp8QuantArgs)
    a.dtype, a.head_dim, a.heads, a.n_tokens = _DT[q.dtype], D, H, S
    a.qk_scale = (1.0 / math.sqrt(D)) if scale is None else scale
    a.q, a.k, a.v = _tensor(q), _tensor(k), _tensor(v)
    a.q8, a.k8, a.v8 = _tensor(out.q), _tensor(out.k), _tensor(out.v)
    a.v_descale, a.ws = out.v_descale.data_ptr(), out.ws.data_ptr()
    if phase not in (None, "stats", "convert"):
        raise ValueError(f"fp8_quantize_qkv: unknown phase {phase!r}")
    if phase is not None and not skip_v:
        raise ValueError("fp8_quantize_qkv: the shard phases convert q and k only (v: fp8_v_absmax / fp8_v_convert)")
    a.flags = (1 if v_per_head else 0) | (2 if center_k else 0) | (4 if skip_v else 0) | \
        (8 if phase == "stats" else 0) | (16 if phase == "convert" else 0)
    a.seg_len, a.tail_first, a.tail_len = seg_len, tail_first, tail_len
    a.video_tokens, a.token_offset, a.total_tokens = video_tokens, token_offset, total_tokens
    if src_map is not None:
        if src_map.dtype != torch.int32 or not src_map.is_cuda or src_map.numel() < H:
            raise ValueError("fp8_quantize_qkv: src_map must be an int32 device tensor with one entry per output head")
        a.src_map = src_map.data_ptr()
    if slots is not None:
        if seg_len <= 0 or not (0 <= slots[0] < slots[1] <= H):
            raise ValueError(f"fp8_quantize_qkv: slots {slots} need the segmented layout and 0 <= first < end <= {H}")
        a.slot_first, a.slot_count = slots[0], slots[1] - slots[0]
    _C.check(_C.lib().vorta_fp8_quantize_qkv(C.byref(a), _stream()), "vorta_fp8_quantize_qkv")
    return out


def fp8_quantize_v(v: torch.Tensor, out: Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]] = None):
    """(H,S,D) bf16 / fp16 view -> (v8 (H,S,D) uint8, v_descale (H,D) float32, amax workspace): the e4m3 copy of v for the
    mixed-precision attention (16-bit scores, e4m3 P V): vorta_fp8_v_absmax + vorta_fp8_v_convert, exact per-(head,
    channel) abs-max.  `out` = a previous result to overwrite."""
    H, S, D = v.shape
    if out is None:
        out = (torch.empty((H, S, D), dtype=FP8_STORAGE, device=v.device),
               torch.empty((H, D), dtype=torch.float32, device=v.device),
               torch.empty((H, D), dtype=torch.float32, device=v.device))
    v8, vd, amax = out
    amax.zero_()
    fp8_v_absmax(v, amax)
    fp8_v_convert(v, amax, v8, v_descale=vd)
    return out


class I8Operands:
    """int8 keys of one layer for the int8-score attention: k8 (same geometry as k, int8), k_bias (one float per row: the
    query centre's product with the row, in units of the head's key scale), q_prep (heads, 2, D): centre and channel
    multipliers the attention kernel applies to its query rows, k_head_scale (heads,), ws: key centre | 1 / multipliers | amax"""
    __slots__ = ("k8", "k_bias", "q_prep", "k_head_scale", "ws")

    def __init__(self, k8, k_bias, q_prep, k_head_scale, ws):
        self.k8, self.k_bias, self.q_prep, self.k_head_scale, self.ws = k8, k_bias, q_prep, k_head_scale, ws

    def k_center(self):
        H, _, D = self.q_prep.shape
        return self.ws[:H * D].view(H, D)

    def heads(self, h0: int, h1: int, k8=None, k_bias=None) -> "I8Operands":
        """the operands of heads [h0, h1) (views), optionally with other k8 / k_bias views of those heads"""
        return I8Operands(self.k8[h0:h1] if k8 is None else k8, self.k_bias[h0:h1] if k_bias is None else k_bias,
                          self.q_prep[h0:h1], self.k_head_scale[h0:h1], self.ws)


def i8_quantize_k(q: torch.Tensor, k: torch.Tensor, *, out: Optional[I8Operands] = None, smooth: bool = True,
                  center: bool = True, heads: Optional[int] = None, seg_len: int = 0, tail_first: int = 0, tail_len: int = 0,
                  slots: Optional[Tuple[int, int]] = None) -> I8Operands:
    """vorta_i8_quantize_k: (H,S,D) bf16 / fp16 views of q (sampled only) and k -> int8 keys (one scale per head), a float
    bias per row, the heads' query centres and channel multipliers (include/vorta_hip.h).  `seg_len > 0`: (1,rows,D) row
    arrays in the Ulysses receive layout, `heads` head slots, tail (text) rows from `tail_first`; `slots` = (first, end):
    only those head slots.  `out` = a previous result to overwrite."""
    _require_gpu(q, k)
    if q.dtype not in _DT or q.dtype != k.dtype or q.shape != k.shape or q.dim() != 3:
        raise ValueError("i8_quantize_k takes (H,S,D) bf16 / fp16 views of q and k of equal shape")
    Hx, S, D = k.shape
    if seg_len > 0:
        if Hx != 1 or not heads or heads < 1:
            raise ValueError("i8_quantize_k: the segmented layout takes (1,rows,D) row arrays and the number of heads")
        H = heads
    else:
        if heads not in (None, Hx):
            raise ValueError("i8_quantize_k: `heads` only applies to the segmented layout")
        H = Hx
    dev = k.device
    if out is None:
        out = I8Operands(torch.empty((Hx, S, D), dtype=torch.int8, device=dev),
                         torch.empty((Hx, S), dtype=torch.float32, device=dev),
                         torch.empty((H, 2, D), dtype=torch.float32, device=dev),
                         torch.empty((H,), dtype=torch.float32, device=dev),
                         torch.empty(2 * H * D + H, dtype=torch.float32, device=dev))
    a = _C.I8QuantArgs()
    a.struct_size = C.sizeof(_C.I8QuantArgs)
    a.dtype, a.head_dim, a.heads, a.n_tokens = _DT[k.dtype], D, H, S
    a.q, a.k, a.k8 = _tensor(q), _tensor(k), _tensor(out.k8)
    a.k_bias, a.k_bias_stride_h = out.k_bias.data_ptr(), out.k_bias.stride(0)
    a.q_prep, a.k_head_scale, a.ws = out.q_prep.data_ptr(), out.k_head_scale.data_ptr(), out.ws.data_ptr()
    a.flags = (0 if smooth else 1) | (0 if center else 2)
    a.seg_len, a.tail_first, a.tail_len = seg_len, tail_first, tail_len
    if slots is not None:
        if seg_len <= 0 or not (0 <= slots[0] < slots[1] <= H):
            raise ValueError(f"i8_quantize_k: slots {slots} need the segmented layout and 0 <= first < end <= {H}")
        a.slot_first, a.slot_count = slots[0], slots[1] - slots[0]
    _C.check(_C.lib().vorta_i8_quantize_k(C.byref(a), _stream()), "vorta_i8_quantize_k")
    return out


# "auto8": a head whose int8 keys have a root mean square below this many counts (the bulk of a heavy-tailed head rounds to
# 0 / +-1 under one scale per head) runs with 16-bit scores instead: at full size Student-t(3) keys sit at 0.3-1.0, the outlier-weight
# families of tests/_fp8_inputs.py (which int8 scores hold) at 4.1-4.8, every other family at 17-25 (profiles/r05_auto8_tail_statistic.txt)
I8_TAIL_MIN_RMS = float(__import__("os").environ.get("VORTA_I8_TAIL_MIN_RMS", "3.2"))


def i8_tail_flags(k8: torch.Tensor, min_rms: Optional[float] = None, out: Optional[torch.Tensor] = None,
                  row_map: Optional[torch.Tensor] = None) -> torch.Tensor:
    """vorta_i8_tail_flags: (H,) int32, 1 for the heads whose int8 keys `k8` (H,rows,D) are too coarse for int8 scores.
    `row_map` (int32, one entry per token): the row of each token inside a head's view (Ulysses receive layout)."""
    _require_gpu(k8)
    if k8.dtype != torch.int8 or k8.dim() != 3:
        raise ValueError("i8_tail_flags takes the (H,S,D) int8 keys of i8_quantize_k")
    H = k8.shape[0]
    n = int(row_map.numel()) if row_map is not None else k8.shape[1]
    if row_map is not None and (row_map.dtype != torch.int32 or not row_map.is_cuda or not row_map.is_contiguous()):
        raise ValueError("i8_tail_flags: row_map must be a contiguous int32 device tensor (one row of the head's view per token)")
    if out is None:
        out = torch.empty((H,), dtype=torch.int32, device=k8.device)
    t = _tensor(k8)
    _C.check(_C.lib().vorta_i8_tail_flags(C.byref(t), H, n, _ptr(row_map), float(I8_TAIL_MIN_RMS if min_rms is None else min_rms),
                                          out.data_ptr(), _stream()), "vorta_i8_tail_flags")
    return out


def split_heads(flags: torch.Tensor, head_list: Optional[torch.Tensor], n_heads: int,
                n_heads_dev: Optional[torch.Tensor] = None):
    """vorta_split_heads: the heads of `head_list[:n]` (None: 0 .. n-1) with flag 0 and with flag 1, order kept, as two
    (head_list, n_heads, n_heads_dev) slot-argument dicts for attn_fwd / coreset_select -- everything stays on the device."""
    _require_gpu(flags)
    dev = flags.device
    lists = torch.zeros((2, max(n_heads, 1)), dtype=torch.int32, device=dev)  # (entries past counts[i] stay a valid head index)
    counts = torch.empty((2,), dtype=torch.int32, device=dev)
    _C.check(_C.lib().vorta_split_heads(head_list.data_ptr() if head_list is not None else None,
                                        n_heads_dev.data_ptr() if n_heads_dev is not None else None, int(n_heads),
                                        flags.data_ptr(), lists[0].data_ptr(), lists[1].data_ptr(), counts.data_ptr(), _stream()),
             "vorta_split_heads")
    return tuple(dict(head_list=lists[i], n_heads=int(n_heads), n_heads_dev=counts[i:i + 1]) for i in range(2))


def _fp8_v_args(v: torch.Tensor, amax: torch.Tensor, per_head: bool):
    _require_gpu(v, amax)
    if v.dtype not in _DT or v.dim() != 3:
        raise ValueError("fp8_v_*: v must be a (H,rows,D) bf16 / fp16 view")
    if amax.dtype != torch.float32 or not amax.is_contiguous() or amax.dim() != 2 or amax.shape[1] != v.shape[2]:
        raise ValueError("fp8_v_*: amax must be a contiguous float32 (heads, D) tensor")
    a = _C.Fp8VArgs()
    a.struct_size = C.sizeof(_C.Fp8VArgs)
    a.dtype, a.head_dim, a.heads, a.n_tokens = _DT[v.dtype], v.shape[2], v.shape[0], v.shape[1]
    a.flags = 1 if per_head else 0
    a.v = _tensor(v)
    a.amax = amax.data_ptr()
    return a


def fp8_v_absmax(v: torch.Tensor, amax: torch.Tensor, *, per_head: bool = False) -> torch.Tensor:
    """vorta_fp8_v_absmax: amax[h][d] = max(amax[h][d], max over rows |v[h][row][d]|) -- the caller zeroes `amax` (H,D)
    first; several calls accumulate.  Under sequence parallelism the ranks all-reduce it with MAX afterwards."""
    if amax.shape[0] < v.shape[0]:
        raise ValueError("fp8_v_absmax: amax has fewer heads than v")
    a = _fp8_v_args(v, amax, per_head)
    _C.check(_C.lib().vorta_fp8_v_absmax(C.byref(a), _stream()), "vorta_fp8_v_absmax")
    return amax


def fp8_v_convert(v: torch.Tensor, amax: torch.Tensor, v8: torch.Tensor, *, src_map: Optional[torch.Tensor] = None,
                  v_descale: Optional[torch.Tensor] = None, per_head: bool = False) -> torch.Tensor:
    """vorta_fp8_v_convert: v8[h] = e4m3(v[src] * 240 / amax[src]), src = src_map[h] (identity without a map); `v_descale`
    (heads of v8, D) receives amax[src] / 240.  v8: (heads, rows, D) uint8 view (any head / row strides that keep rows
    16-byte aligned)."""
    a = _fp8_v_args(v, amax, per_head)
    _require_gpu(v8, src_map, v_descale)
    if v8.dtype != FP8_STORAGE or v8.dim() != 3 or v8.shape[1:] != v.shape[1:]:
        raise ValueError("fp8_v_convert: v8 must be a (heads, rows, D) uint8 view with v's rows and channels")
    a.heads = v8.shape[0]
    a.v8 = _tensor(v8)
    if src_map is not None:
        if src_map.dtype != torch.int32 or not src_map.is_contiguous() or src_map.numel() != v8.shape[0]:
            raise ValueError("fp8_v_convert: src_map must be a contiguous int32 tensor with one entry per head of v8")
        a.src_map = src_map.data_ptr()
    elif v8.shape[0] != v.shape[0]:
        raise ValueError("fp8_v_convert: without a source map v8 needs as many heads as v")
    if v_descale is not None:
        if v_descale.dtype != torch.float32 or not v_descale.is_contiguous() or tuple(v_descale.shape) != (v8.shape[0], v.shape[2]):
            raise ValueError("fp8_v_convert: v_descale must be a contiguous float32 (heads of v8, D) tensor")
        a.v_descale = v_descale.data_ptr()
    _C.check(_C.lib().vorta_fp8_v_convert(C.byref(a), _stream()), "vorta_fp8_v_convert")
    return v8


def coreset_select(x: torch.Tensor, latent: Sequence[int], group: Sequence[int], n_keep: int, *,
                   head_list: Optional[torch.Tensor] = None, n_heads: Optional[int] = None,
                   n_heads_dev: Optional[torch.Tensor] = None, tail_first: int = 0, n_tail: int = 0,
                   row_map: Optional[torch.Tensor] = None, want_drop: bool = True, want_keep: bool = True,
                   want_kv: bool = False):
    """vorta_coreset_select: returns keep_rows (slots, G*(1+n_keep)+n_tail) [packed: centres, kept margins, tail] and
    drop_rows (slots, G, g-1-n_keep); with `want_kv` a third tensor, the same rows in group-major ascending order (the
    key-side list: order is free there and neighbours in memory stay neighbours in the list)."""
    _require_gpu(x)
    if n_heads is None:
        n_heads = head_list.numel() if head_list is not None else x.shape[0]
    g = group[0] * group[1] * group[2]
    G = (latent[0] // group[0]) * (latent[1] // group[1]) * (latent[2] // group[2])
    n_list = G * (1 + n_keep) + n_tail
    keep = torch.empty((n_heads, n_list), dtype=torch.int32, device=x.device) if want_keep else None
    kv = torch.empty((n_heads, n_list), dtype=torch.int32, device=x.device) if want_kv else None
    drop = torch.empty((n_heads, G, g - 1 - n_keep), dtype=torch.int32, device=x.device) if want_drop else None
    a = _C.CoresetArgs()
    a.struct_size = C.sizeof(_C.CoresetArgs)
    a.dtype, a.head_dim, a.n_heads = _DT[x.dtype], x.shape[-1], n_heads
    a.x = _tensor(x)
    a.head_list, a.n_heads_dev = _ptr(head_list), _ptr(n_heads_dev)
    a.latent = (C.c_int32 * 3)(*latent)
    a.group = (C.c_int32 * 3)(*group)
    a.n_keep, a.tail_first, a.n_tail = n_keep, tail_first, n_tail
    a.row_map = _ptr(row_map)
    if keep is not None:
        a.keep_rows, a.keep_rows_stride_h = keep.data_ptr(), keep.stride(0)
    if kv is not None:
        a.keep_rows_kv, a.keep_rows_kv_stride_h = kv.data_ptr(), kv.stride(0)
    if drop is not None:
        a.drop_rows, a.drop_rows_stride_h = drop.data_ptr(), drop.stride(0)
    _C.check(_C.lib().vorta_coreset_select(C.byref(a), _stream()), "vorta_coreset_select")
    return (keep, drop, kv) if want_kv else (keep, drop)


def sta_table_sizes(latent, tile, window, t_eff: int = 0) -> Tuple[int, int, int]:
    """(n_tiles, tokens_per_tile, n_kv) -- host-only geometry query (works without a GPU)."""
    a = _C.StaArgs()
    a.struct_size = C.sizeof(_C.StaArgs)
    a.latent, a.tile, a.window = (C.c_int32 * 3)(*latent), (C.c_int32 * 3)(*tile), (C.c_int32 * 3)(*window)
    a.t_eff = t_eff
    nt, tok, nkv = C.c_int32(), C.c_int32(), C.c_int32()
    _C.check(_C.lib().vorta_sta_table_sizes(C.byref(a), C.byref(nt), C.byref(tok), C.byref(nkv)), "vorta_sta_table_sizes")
    return nt.value, tok.value, nkv.value


def sta_build_tables(latent, tile, window, t_eff: int, device, row_map: Optional[torch.Tensor] = None
                     ) -> Tuple[torch.Tensor, torch.Tensor]:
    """vorta_sta_build_tables: q_rows (S,) and kv_rows (n_tiles, n_kv) int32 on `device`."""
    n_tiles, tok, n_kv = sta_table_sizes(latent, tile, window, t_eff)
    S = latent[0] * latent[1] * latent[2]
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _C.VortaHipError("sta_build_tables needs a cuda device")
    q_rows = torch.empty(S, dtype=torch.int32, device=dev)
    kv_rows = torch.empty((n_tiles, n_kv), dtype=torch.int32, device=dev)
    a = _C.StaArgs()
    a.struct_size = C.sizeof(_C.StaArgs)
    a.latent, a.tile, a.window = (C.c_int32 * 3)(*latent), (C.c_int32 * 3)(*tile), (C.c_int32 * 3)(*window)
    a.t_eff = t_eff
    a.row_map = _ptr(row_map)
    a.q_rows, a.kv_rows = q_rows.data_ptr(), kv_rows.data_ptr()
    with torch.cuda.device(dev):
        _C.check(_C.lib().vorta_sta_build_tables(C.byref(a), _stream()), "vorta_sta_build_tables")
    return q_rows, kv_rows


def router_route(temb: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, heads: int, tau: float,
                 n_experts: int = 3):
    """vorta_router_route: returns (scores (B,H,E) in temb.dtype, expert_of_head (H,), head_lists (E,H), head_counts (E,))
    -- all on device; nothing is synchronised."""
    _require_gpu(temb, weight, bias)
    B, E = temb.shape
    dev = temb.device
    temb, weight, bias = temb.contiguous(), weight.contiguous(), bias.contiguous()
    scores = torch.empty((B, heads, n_experts), dtype=temb.dtype, device=dev)
    expert = torch.empty(heads, dtype=torch.int32, device=dev)
    lists = torch.zeros((n_experts, heads), dtype=torch.int32, device=dev)
    counts = torch.empty(n_experts, dtype=torch.int32, device=dev)
    ws = torch.empty(B * heads * n_experts, dtype=torch.float32, device=dev)
    a = _C.RouterArgs()
    a.struct_size = C.sizeof(_C.RouterArgs)
    a.dtype = _DT[temb.dtype]
    a.batch, a.embed_dim, a.heads, a.n_experts = B, E, heads, n_experts
    a.temb, a.weight, a.bias = temb.data_ptr(), weight.data_ptr(), bias.data_ptr()
    a.tau = tau
    a.scores, a.expert_of_head = scores.data_ptr(), expert.data_ptr()
    a.head_lists, a.head_counts, a.ws_logits = lists.data_ptr(), counts.data_ptr(), ws.data_ptr()
    _C.check(_C.lib().vorta_router_route(C.byref(a), _stream()), "vorta_router_route")
    return scores, expert, lists, counts


def route_plan(temb: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, heads: int, tau: float,
               n_experts: int = 3, out=None):
    """vorta_route_plan: the routes of ALL layers from one timestep embedding.
    temb (B,E); weight (L,heads*n_experts,E); bias (L,heads*n_experts).  Returns
    (scores (L,B,H,E'), expert_of_head (L,H), head_lists (L,E',H), head_counts (L,E')), all on device, no sync.
    `out` = a previous result to overwrite in place (fixed addresses: hipGraph replay)."""
    _require_gpu(temb, weight, bias)
    B, E = temb.shape
    L = weight.shape[0]
    if weight.shape != (L, heads * n_experts, E) or bias.shape != (L, heads * n_experts):
        raise ValueError(f"route_plan: weight {tuple(weight.shape)} / bias {tuple(bias.shape)} do not match "
                         f"(L, {heads * n_experts}, {E})")
    dev = temb.device
    temb, weight, bias = temb.contiguous(), weight.contiguous(), bias.contiguous()
    if out is None:
        scores = torch.empty((L, B, heads, n_experts), dtype=temb.dtype, device=dev)
        expert = torch.empty((L, heads), dtype=torch.int32, device=dev)
        lists = torch.zeros((L, n_experts, heads), dtype=torch.int32, device=dev)
        counts = torch.empty((L, n_experts), dtype=torch.int32, device=dev)
    else:
        scores, expert, lists, counts = out
    ws = torch.empty(L * B * heads * n_experts, dtype=torch.float32, device=dev)
    a = _C.RouterArgs()
    a.struct_size = C.sizeof(_C.RouterArgs)
    a.dtype = _DT[temb.dtype]
    a.batch, a.embed_dim, a.heads, a.n_experts = B, E, heads, n_experts
    a.temb, a.weight, a.bias = temb.data_ptr(), weight.data_ptr(), bias.data_ptr()
    a.tau = tau
    a.scores, a.expert_of_head = scores.data_ptr(), expert.data_ptr()
    a.head_lists, a.head_counts, a.ws_logits = lists.data_ptr(), counts.data_ptr(), ws.data_ptr()
    _C.check(_C.lib().vorta_route_plan(C.byref(a), L, _stream()), "vorta_route_plan")
    return scores, expert, lists, counts


def route_scores(scores: torch.Tensor, tau: float):
    """vorta_route_scores: top-1 / tau dispatch of an existing (B,H,E) score tensor (batch item 0 routes).
    Returns (expert_of_head (H,), head_lists (E,H), head_counts (E,)) on device, no host sync."""
    _require_gpu(scores)
    if scores.dtype not in _DT and scores.dtype != torch.float32:
        scores = scores.float()
    scores = scores.contiguous()
    B, H, E = scores.shape
    dev = scores.device
    expert = torch.empty(H, dtype=torch.int32, device=dev)
    lists = torch.zeros((E, H), dtype=torch.int32, device=dev)
    counts = torch.empty(E, dtype=torch.int32, device=dev)
    a = _C.RouterArgs()
    a.struct_size = C.sizeof(_C.RouterArgs)
    a.dtype = _DT.get(scores.dtype, 2)  # 2 = VORTA_FP32 (accepted by vorta_route_scores only)
    a.batch, a.embed_dim, a.heads, a.n_experts = B, 0, H, E
    a.tau = tau
    a.scores, a.expert_of_head = scores.data_ptr(), expert.data_ptr()
    a.head_lists, a.head_counts = lists.data_ptr(), counts.data_ptr()
    _C.check(_C.lib().vorta_route_scores(C.byref(a), _stream()), "vorta_route_scores")
    return expert, lists, counts


def qk_norm_rope(x: torch.Tensor, weight: Optional[torch.Tensor], eps: float, *,
                 cos: Optional[torch.Tensor] = None, sin: Optional[torch.Tensor] = None,
                 n_tokens: Optional[int] = None, token_offset: int = 0, rope_tokens: Optional[int] = None,
                 across_heads: bool = False) -> torch.Tensor:
    """vorta_qk_norm_rope: in-place RMSNorm (+ rotary embedding) of a (H,S,D) view of q or k."""
    _require_gpu(x, weight, cos, sin)
    if x.dtype not in _DT:
        raise ValueError("x must be bf16 or fp16")
    a = _C.NormRopeArgs()
    a.struct_size = C.sizeof(_C.NormRopeArgs)
    a.dtype, a.head_dim, a.heads = _DT[x.dtype], x.shape[-1], x.shape[0]
    a.x = _tensor(x)
    if weight is not None:
        if weight.dtype != x.dtype or not weight.is_contiguous():
            weight = weight.to(x.dtype).contiguous()
        a.weight = weight.data_ptr()
    n_tokens = x.shape[1] - token_offset if n_tokens is None else n_tokens
    if cos is not None:
        if cos.dtype != torch.float32 or not cos.is_contiguous():
            cos = cos.float().contiguous()
        if sin.dtype != torch.float32 or not sin.is_contiguous():
            sin = sin.float().contiguous()
        a.cos, a.sin = cos.data_ptr(), sin.data_ptr()
        rope_tokens = min(cos.shape[0], n_tokens) if rope_tokens is None else rope_tokens
        if cos.shape[0] < rope_tokens or cos.shape[-1] != x.shape[-1]:
            raise ValueError("cos/sin must be (>= rope_tokens, D)")
    a.n_tokens, a.token_offset, a.rope_tokens = n_tokens, token_offset, rope_tokens or 0
    a.eps = eps
    a.across_heads = 1 if across_heads else 0
    _C.check(_C.lib().vorta_qk_norm_rope(C.byref(a), _stream()), "vorta_qk_norm_rope")
    return x


def mix_experts(xs, scores: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """vorta_mix_experts: out[h] = sum_e scores[0,h,e] * xs[e][h] for (H,N,D) views (hunyuan.py:509-513)."""
    _require_gpu(scores, out, *xs)
    if len(xs) != 3 or scores.shape[-1] != 3:
        raise ValueError("mix_experts takes the outputs of the three experts and (B,H,3) scores")
    H, N, D = out.shape
    sc = scores[0].to(out.dtype).contiguous()
    a = _C.MixArgs()
    a.struct_size = C.sizeof(_C.MixArgs)
    a.dtype, a.head_dim, a.heads, a.n_experts, a.n_rows = _DT[out.dtype], D, H, 3, N
    for e in range(3):
        if xs[e].shape != out.shape or xs[e].dtype != out.dtype:
            raise ValueError("mix_experts: expert outputs must match the output's shape and dtype")
        a.x[e] = _tensor(xs[e])
    a.out = _tensor(out)
    a.scores = sc.data_ptr()
    _C.check(_C.lib().vorta_mix_experts(C.byref(a), _stream()), "vorta_mix_experts")
    return out


def permute_heads(srcs, dsts, src_map: Optional[torch.Tensor] = None, dst_map: Optional[torch.Tensor] = None) -> None:
    """vorta_permute_heads: dsts[t][dst_map[h]] = srcs[t][src_map[h]] for up to four (H,N,D) views in one launch (the
    staging passes of the Ulysses exchange, vorta/ulysses/utils.py:61-91).  Maps: int32 device tensors of H entries."""
    _require_gpu(*srcs, *dsts)
    if not srcs or len(srcs) != len(dsts) or len(srcs) > 4:
        raise ValueError("permute_heads takes one to four source / destination pairs")
    H, N, D = dsts[0].shape
    a = _C.PermuteArgs()
    a.struct_size = C.sizeof(_C.PermuteArgs)
    dt = _C.VORTA_FP8E4M3 if dsts[0].dtype == FP8_STORAGE else _DT[dsts[0].dtype]
    a.dtype, a.head_dim, a.heads, a.n_rows, a.n_tensors = dt, D, H, N, len(srcs)
    for t, (x, y) in enumerate(zip(srcs, dsts)):
        if x.dtype != y.dtype or y.dtype != dsts[0].dtype or tuple(y.shape) != (H, N, D) or x.shape[1:] != y.shape[1:]:
            raise ValueError("permute_heads: every pair must be (heads, rows, D) views of one dtype and row count")
        if x.stride(-1) != 1 or y.stride(-1) != 1:
            raise ValueError("permute_heads: the channel dimension must be contiguous")
        a.src[t], a.dst[t] = _tensor(x), _tensor(y)
    for name, m, n_src in (("src_map", src_map, srcs[0].shape[0]), ("dst_map", dst_map, H)):
        if m is not None:
            if m.dtype != torch.int32 or m.numel() != H or not m.is_contiguous() or m.device != dsts[0].device:
                raise ValueError(f"permute_heads: {name} must be a contiguous int32 device tensor with one entry per head")
            setattr(a, name, m.data_ptr())
    if src_map is None and srcs[0].shape[0] != H:
        raise ValueError("permute_heads: without a source map the sources need as many heads as the destinations")
    _C.check(_C.lib().vorta_permute_heads(C.byref(a), _stream()), "vorta_permute_heads")


def seq_row_map(n_tokens: int, seg_len: int, seg_stride_rows: int, device) -> torch.Tensor:
    """vorta_seq_row_map (zero-copy Ulysses layout)."""
    out = torch.empty(n_tokens, dtype=torch.int32, device=device)
    with torch.cuda.device(out.device):
        _C.check(_C.lib().vorta_seq_row_map(out.data_ptr(), n_tokens, seg_len, seg_stride_rows, _stream()),
                 "vorta_seq_row_map")
    return out
