"""Zero-copy Ulysses sequence parallelism for the routed attention op (MI355X: RCCL all-to-all over xGMI).

Reference flow per tensor (vorta/ulysses/utils.py:61-91): transpose+contiguous -> all_to_all_single ->
device sync -> transpose+contiguous, i.e. four extra HBM passes per tensor and a host stall, applied per
routed subset (hunyuan.py:153-155,425-427,483-485), which also forces h_e % P == 0.

Here the heads of the local sequence shard bound for one peer travel as ONE contiguous (H/P, S/P, D) message
into their final place in the receiver's buffer, and the attention kernels read that buffer in place through
a row table (`row_map`): token s of local head slot i lives at row
    (s // Sl) * (Hl*Sl) + i*Sl + s % Sl                      (Sl = S/P, Hl = H/P)
so the receive side has no unpack pass (the send side orders the heads with one gather pass, or none when they
already are).  The output travels back the same way.  All H heads are resharded once, BEFORE routing (any expert
mix works), and the head -> rank placement is free: `balanced_head_order` gives every rank the same number of heads and a near-equal sum
of expert costs (routes depend only on the timestep, so they are known when the layer starts).
Text tokens are replicated: each rank copies its heads' text rows behind the video rows
(`shrink_dim` in the reference, hunyuan.py:158-160) and the text outputs are all-gathered over heads (:187).
The transport is ONE `all_to_all_single` per tensor -- the reference's collective (utils.py:48,80) -- for the whole layer, or per
slot group when the exchange is overlapped with the attention: a slot group is a receive layout of its own inside the same
buffers (`UlyssesLayout.grouping`), with its own row map.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from .. import ops


# Transport of the head exchange: ONE `all_to_all_single` per tensor and slot group (the collective the reference uses,
# vorta/ulysses/utils.py:48,80) -- a slot group is a receive layout of its own (`UlyssesLayout.grouping`), so its chunks are
# contiguous on both sides like the whole-tensor exchange's.  There is no second transport (rounds 2-5 kept grouped send / recv
# for the slot groups: a branch the one-rank RCCL rehearsal could not reach).
from .._debug import flag as _debug_flag  # A/B switches: VORTA_DEBUG="key=value,..." (vorta_amd/_debug.py)
# staging passes of device tensors: one vorta_permute_heads launch each ("hip"); sp_staging=torch keeps the index ops the CPU
# rehearsals use (A/B measurements only)
HIP_STAGING = _debug_flag("sp_staging", "hip") != "torch"
# slot groups attend on alternating HIP streams (sp_group_streams=0: all on the current stream, A/B)
GROUP_STREAMS = _debug_flag("sp_group_streams", "1") != "0"
# One-GPU emulation of a rank (loopback) with a WIRE: every collective of the exchange holds a side stream for the time its
# largest chunk needs at this many GB/s per xGMI link (+ 10 us), and `_finish` makes the consumer wait for it -- so the
# emulation ranks 1 / 2 / 3 slot groups by what the overlap hides.  0 (default) = transfers are free, as in rounds 2-4.
# An ASSUMPTION about the links, not a measurement of them (profiles/r05_sp_groups_emulated.txt).
EMULATE_LINK_GBPS = float(_debug_flag("sp_emulate_link_gbps", "0") or 0)
# Rehearsal on a world of ONE rank (tests/test_hip_rccl_single_rank.py): issue the collectives of the whole-tensor exchange --
# all_to_all_single with one chunk, the all-reduces, the text all-gather -- although there is no peer, so that the direct RCCL
# branch runs on the one-GPU box (RCCL refuses two ranks on one device).  Off in every product run.
FORCE_COLLECTIVES = __import__("os").environ.get("VORTA_SP_FORCE_COLLECTIVES", "0") == "1"
_WIRE = {}


def _wire_sleep(device, nbytes_per_link: float):
    """handle whose `_finish` waits until a transfer of `nbytes_per_link` over one link WOULD have ended: a spin kernel of that
    duration on the wire stream (one thread: it takes no CU from the attention), ordered after the current stream"""
    idx = torch.device(device).index or 0
    w = _WIRE.get(idx)
    if w is None:
        st = torch.cuda.Stream(device=device)
        # cycles of torch.cuda._sleep per microsecond, measured once
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(st):
            e0.record(st)
            torch.cuda._sleep(20_000_000)
            e1.record(st)
        e1.synchronize()
        w = _WIRE[idx] = (st, 20_000_000 / (e0.elapsed_time(e1) * 1e3))
    st, cyc_per_us = w
    us = 10.0 + nbytes_per_link / (EMULATE_LINK_GBPS * 1e3)
    st.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(st):
        torch.cuda._sleep(int(us * cyc_per_us))
        ev = torch.cuda.Event()
        ev.record(st)
    return ("event", ev)


def group_sizes(Hl: int, groups: int) -> List[int]:
    """sizes of `groups` slot groups of Hl local heads: as equal as possible, the larger ones first (5 -> 3 + 2)"""
    groups = max(1, min(int(groups), Hl))
    base, rem = divmod(Hl, groups)
    return [base + 1] * rem + [base] * (groups - rem)


def balanced_head_order(experts: Sequence[int], cost_of_expert: Sequence[float], P: int, groups: int = 1) -> List[int]:
    """Heads grouped by destination rank (rank j owns order[j*Hl:(j+1)*Hl]): exactly Hl heads per rank,
    greedy longest-processing-time on the expert costs.  Inside a rank the heads are ascending (groups = 1) or,
    for the overlapped exchange, split the same way into `groups` slot groups (`group_sizes`) of near-equal cost per head
    slot (ascending inside a group).  Deterministic, so every rank computes the same order without communicating."""
    H = len(experts)
    assert H % P == 0, f"heads {H} must be divisible by the sequence-parallel size {P}"
    Hl = H // P

    def lpt(heads, sizes):
        load = [0.0] * len(sizes)
        bins: List[List[int]] = [[] for _ in sizes]
        for h in sorted(heads, key=lambda i: (-cost_of_expert[int(experts[i])], i)):
            # fill by load per slot, so a larger group takes proportionally more work
            j = min((r for r in range(len(sizes)) if len(bins[r]) < sizes[r]), key=lambda r: (load[r] / sizes[r], r))
            bins[j].append(h)
            load[j] += cost_of_expert[int(experts[h])]
        return bins

    order: List[int] = []
    sizes = group_sizes(Hl, groups)
    for b in lpt(range(H), [Hl] * P):
        if len(sizes) == 1:
            order += sorted(b)
        else:
            for g in lpt(b, sizes):
                order += sorted(g)
    return order


def balanced_placement(experts: Sequence[int], cost_of_expert: Sequence[float], P: int, groups: int = 1,
                       max_heads: Optional[int] = None):
    """Like `balanced_head_order`, but ranks may hold different NUMBERS of heads: whole heads are a coarse unit when H/P is
    small and the experts' costs differ by 4-5 x (Wan-14B at P = 8: five heads per rank, 14 full-attention heads -> the
    heaviest rank carries 2 full + 3 sliding-tile heads, 1.04 of the mean; a trained router's mixes are less even than
    that).  Greedy longest-processing-time with no equal-count constraint: every head goes to the least-loaded rank (at
    most `max_heads` per rank, default 2 H/P; at least one head everywhere).  Returns (order, counts): rank j owns
    order[sum(counts[:j]) : sum(counts[:j + 1])]; inside a rank as `balanced_head_order`.  Deterministic."""
    H = len(experts)
    assert H >= P, f"{H} heads cannot cover {P} ranks"
    cap = max_heads if max_heads is not None else max(2, 2 * ((H + P - 1) // P))
    assert cap * P >= H
    load = [0.0] * P
    bins: List[List[int]] = [[] for _ in range(P)]
    for h in sorted(range(H), key=lambda i: (-cost_of_expert[int(experts[i])], i)):
        # an empty rank first (every rank ends with a head), then the least loaded one with room
        j = min((r for r in range(P) if len(bins[r]) < cap), key=lambda r: (len(bins[r]) > 0, load[r], r))
        bins[j].append(h)
        load[j] += cost_of_expert[int(experts[h])]
    counts = [len(b) for b in bins]
    n_groups = max(1, min(int(groups), min(counts)))
    order: List[int] = []
    for b in bins:
        if n_groups == 1:
            order += sorted(b)
            continue
        sizes = group_sizes(len(b), n_groups)
        gl = [0.0] * n_groups
        gb: List[List[int]] = [[] for _ in sizes]
        for h in sorted(b, key=lambda i: (-cost_of_expert[int(experts[i])], i)):
            g = min((r for r in range(n_groups) if len(gb[r]) < sizes[r]), key=lambda r: (gl[r] / sizes[r], r))
            gb[g].append(h)
            gl[g] += cost_of_expert[int(experts[h])]
        for g in gb:
            order += sorted(g)
    return order, counts


def split_align(n_tokens: int) -> int:
    """granularity of a query-range boundary: the 256 query rows of a workgroup (a part then starts on a workgroup of the
    whole-head launch); 32 = one wave only where a head has a handful of workgroups (tests, tiny shapes)"""
    return 256 if n_tokens >= 16 * 256 else 32


def split_placement(experts: Sequence[int], cost_of_expert: Sequence[float], P: int, n_tokens: int, groups: int = 1,
                    max_heads: Optional[int] = None, tol: float = 0.01, align: int = 256, max_parts: int = 2):
    """`balanced_placement`, then BELOW whole heads: while the heaviest rank carries more than (1 + tol) x the mean, one of
    its full-attention heads gives the tail of its QUERIES to the lightest rank -- both ranks receive the head's K and V
    (and q: the exchange stays one message per peer), each computes its own query range and returns the whole head; the
    token shards pick their rows from the rank that computed them.  Full attention costs the same per query, so a
    range moves load continuously (`align`-token steps): water-filling towards the mean, heaviest rank first.  Wan-14B at
    P = 8 (five heads per rank, costs 5.6 : 1.4 : 1) goes from 1.037 of the mean to <= 1.01.  A rank holds at most
    `max_parts` partial heads (each is one more segment of its fused launch).  Returns (order, counts, parts): the head
    order lists a split head once per part -- rank j owns order[sum(counts[:j]) : sum(counts[:j + 1])], sum(counts) =
    H + number of extra parts -- and parts[i] = None for a whole head or (t0, t1), the VIDEO query tokens slot i computes;
    the part that ends at the last video token (t1 == n_tokens) also answers the head's text queries, which follow the
    video in token order.  Deterministic."""
    if align % 32 or not 1 <= max_parts <= 2:
        # a wave = 32 consecutive query positions shares reference-point decisions (and, with int8 scores, its query scale):
        # ranges on 32-token boundaries keep every wave's rows together, so a split head is the whole head bit for bit
        raise ValueError("split_placement: `align` must be a multiple of 32 tokens, `max_parts` 1 or 2")
    order, counts = balanced_placement(experts, cost_of_expert, P, 1, max_heads)
    starts = [sum(counts[:j]) for j in range(P + 1)]
    bins = [list(order[starts[j]:starts[j + 1]]) for j in range(P)]
    cost = lambda h: cost_of_expert[int(experts[h])]
    full = max(range(len(cost_of_expert)), key=lambda e: cost_of_expert[e])
    c_full = cost_of_expert[full]
    load = [sum(cost(h) for h in b) for b in bins]
    mean = sum(load) / P
    rng = {}                      # (rank, head) -> [t0, t1): the query range of a partial head
    n_parts = [0] * P             # partial heads per rank
    step = c_full * align / n_tokens
    for _ in range(4 * P):
        R = max(range(P), key=lambda j: (load[j], -j))
        if load[R] <= (1.0 + tol) * mean:
            break
        # a full-attention head of R that can still give: one it already holds in part, else (room permitting) a whole one
        cand = [h for h in bins[R] if int(experts[h]) == full and (((R, h) in rng) or n_parts[R] < max_parts)]
        cand = [h for h in cand if rng.get((R, h), (0, n_tokens))[1] - rng.get((R, h), (0, n_tokens))[0] > 2 * align]
        light = [j for j in range(P) if j != R and n_parts[j] < max_parts and load[j] < mean - step
                 and not any((j, h) in rng for h in cand)]
        if not cand or not light:
            break
        r = min(light, key=lambda j: (load[j], j))
        h = max(cand, key=lambda x: (rng.get((R, x), (0, n_tokens))[1] - rng.get((R, x), (0, n_tokens))[0], x))
        t0, t1 = rng.get((R, h), (0, n_tokens))
        move = min(load[R] - mean, mean - load[r])
        n = min(int(round(move / step)) * align, (t1 - t0) - align)
        # the boundary is counted from the FRONT of the sequence in `align` steps: only the part that ends at the last
        # token may be ragged (n_tokens itself need not be a multiple of 32: 32 760, 75 600), so every wave and every
        # 256-row workgroup of a part is a wave / workgroup of the whole-head launch
        cut = ((t1 - n) // align) * align
        if cut * 1 <= t0 or t1 - cut < align or cut - t0 < align:
            break
        n = t1 - cut
        if (R, h) not in rng:
            n_parts[R] += 1
        rng[(R, h)] = (t0, cut)
        rng[(r, h)] = (cut, t1)
        n_parts[r] += 1
        bins[r].append(h)
        load[R] -= n / align * step
        load[r] += n / align * step
    eff = lambda j, h: cost(h) * ((rng[(j, h)][1] - rng[(j, h)][0]) / n_tokens if (j, h) in rng else 1.0)
    counts = [len(b) for b in bins]
    n_groups = max(1, min(int(groups), min(counts)))
    out_order: List[int] = []
    parts: List[Optional[tuple]] = []
    for j, b in enumerate(bins):
        if n_groups == 1:
            slots = sorted(b)
        else:
            sizes = group_sizes(len(b), n_groups)
            gl = [0.0] * n_groups
            gb: List[List[int]] = [[] for _ in sizes]
            for h in sorted(b, key=lambda i: (-eff(j, i), i)):
                g = min((x for x in range(n_groups) if len(gb[x]) < sizes[x]), key=lambda x: (gl[x] / sizes[x], x))
                gb[g].append(h)
                gl[g] += eff(j, h)
            slots = [h for g in gb for h in sorted(g)]
        out_order += slots
        parts += [rng.get((j, h)) for h in slots]
    return out_order, counts, parts


def placement_loads(experts: Sequence[int], cost_of_expert: Sequence[float], order: Sequence[int], counts: Sequence[int],
                    parts: Optional[Sequence[Optional[tuple]]] = None, n_tokens: int = 1) -> List[float]:
    """cost per rank of a placement (a partial full-attention head counts by its share of the queries)"""
    starts = [sum(counts[:j]) for j in range(len(counts) + 1)]
    loads = []
    for j in range(len(counts)):
        tot = 0.0
        for i in range(starts[j], starts[j + 1]):
            c = cost_of_expert[int(experts[order[i]])]
            if parts is not None and parts[i] is not None:
                c *= (parts[i][1] - parts[i][0]) / n_tokens
            tot += c
        loads.append(tot)
    return loads


def slot_groups(Hl: int, groups: int) -> List[tuple]:
    out, g0 = [], 0
    for n in group_sizes(Hl, groups):
        out.append((g0, g0 + n))
        g0 += n
    return out


def exchange_and_attend(lay: "UlyssesLayout", shards, bufs, head_order, texts, groups, attend, out_shard, out_text,
                        vwire: Optional["VWire"] = None, prepare=None, parts=None):
    """One layer under sequence parallelism.  `groups` = slot ranges of the local heads; `attend(g0, g1, index)`
    enqueues the attention over local head slots [g0, g1) of the layout buffers.  With one group this is
    scatter -> attention -> gather.  With several, the exchange of group g+1 and the return of group g-1 are in
    flight while group g computes (both run on the communicator's stream, the attention on the current one).
    With several groups the receive buffers are GROUP-MAJOR: slot group gi is a receive layout of its own inside them
    (`lay.grouping(len(groups))[0][gi]`: `.lay` the sub-layout with its row map, `.buffer(buf)` its rows, `.head_view(buf)`),
    which is what `attend` reads and writes -- so every group's exchange is one all_to_all_single per tensor.
    `vwire`: v travels as e4m3 (converted on this side with the scales of the whole sequence) into `vwire.buf`.
    `prepare()`: builds, on the current stream and BEFORE the group streams fork, every cached device object `attend` would
    otherwise build lazily (geometry tables, routing lists): a table built inside group 0's stream is not ordered against
    group 1's first read of the cached object on the other stream."""
    if prepare is not None:
        prepare()
    handles = lay.scatter_heads_start(shards, bufs[:3], head_order, texts, groups, vwire=vwire)
    state = lay.gather_heads_begin(out_shard, head_order, parts, n_groups=len(groups))
    back = []
    side = None
    if len(groups) > 1 and GROUP_STREAMS and out_shard.is_cuda:
        # consecutive slot groups attend on alternating streams: a group's launch is small (H/P/groups heads), and on
        # one stream every group would pay its own tail -- measured on a rank of 8 (Wan-14B fp8, no transfers) 294 ms
        # with one group, 347 with two, 453 with five; with the next group already resident its workgroups fill the
        # tail.  Each group's stream waits for ITS exchange only.
        cur = torch.cuda.current_stream(out_shard.device)
        side = [cur] + _group_streams(out_shard.device)
        fork = torch.cuda.Event()
        fork.record(cur)
        for st in side[1:]:
            st.wait_event(fork)
    for gi, (g0, g1) in enumerate(groups):
        if side is None:
            lay._finish(handles[gi])
            attend(g0, g1, gi)
            back.append(lay.gather_heads_start(bufs[3], state, (g0, g1), gi, len(groups)))
        else:
            with torch.cuda.stream(side[gi % len(side)]):
                lay._finish(handles[gi])
                attend(g0, g1, gi)
                back.append(lay.gather_heads_start(bufs[3], state, (g0, g1), gi, len(groups)))
    if side is not None:  # join: everything below (and the caller) is ordered after every group
        for st in side[1:]:
            done = torch.cuda.Event()
            done.record(st)
            cur.wait_event(done)
    for h in back:
        lay._finish(h)
    lay.gather_heads_end(bufs[3], state, out_text)


def tag_rows(t: int, heads: Sequence[int], tokens: torch.Tensor, kind: int, D: int, dtype, device) -> torch.Tensor:
    """(len(heads), len(tokens), D) rows whose VALUES name where they belong: channel 0 = head + 1, channels 1-3 = the token
    index in base 128, channel 4 = tensor id + 1 (q, k, v), channel 5 = `kind` (1 video row, 2 text row), the rest a hash
    of (token, head, channel, tensor) in [-125, 125].  Small integers: exact in bf16 and fp16, so a row that travelled
    through the exchange can be compared for EQUALITY with the row that should have arrived."""
    h = torch.as_tensor(list(heads), dtype=torch.int64, device=device).view(-1, 1, 1)
    s = tokens.to(device=device, dtype=torch.int64).view(1, -1, 1)
    c = torch.arange(D, dtype=torch.int64, device=device).view(1, 1, -1)
    x = (s * 31 + h * 17 + c * 7 + t * 5) % 251 - 125
    x[..., 0] = (h + 1).expand(-1, s.shape[1], 1)[..., 0]
    x[..., 1] = (s & 127).expand(h.shape[0], -1, 1)[..., 0]
    x[..., 2] = ((s >> 7) & 127).expand(h.shape[0], -1, 1)[..., 0]
    x[..., 3] = ((s >> 14) & 127).expand(h.shape[0], -1, 1)[..., 0]
    x[..., 4] = t + 1
    x[..., 5] = kind
    return x.to(dtype)


def exchange_selfcheck(lay: "UlyssesLayout", head_order: Sequence[int], groups, bufs, vwire: Optional["VWire"] = None,
                       break_order: bool = False, parts=None) -> dict:
    """Push integer-tagged q, k, v (`tag_rows`: value = f(tensor, head, token, channel)) through THIS layout's own exchange
    -- the staging pass, the all_to_all_single (even or per-rank splits) of every slot group, the text rows, v as e4m3 when it travels that way, the return trip of the output and the all-gather of the text
    outputs -- with identity in place of attention (o = q), and compare EXACTLY on every rank:
      * every row of every local head slot of the q, k, v receive buffers, read through `row_map` as the kernels read
        them, against the row that head and token should hold (v as e4m3: against the bytes and scales this rank computes
        from the whole tagged sequence of its heads: abs-max over shards = abs-max over the sequence);
      * the returned sequence shard of every head and the gathered text rows against the q rows sent.
    Replaces nothing in the reference (vorta/ulysses/utils.py:42-56,68-89 has no check); it is what makes the first run
    on a new transport trustworthy.  `break_order` (tests only): the last rank swaps two heads of ITS copy of the order,
    which must fail the check.  `parts` (`split_placement`): a slot that computes only a range of its head's queries
    returns ZEROS for the other rows, so the shard is whole only if every row was taken from the right slot.  Returns {"ok", "failed": [names], "bytes": sent + received by this rank, "ms"}."""
    import time
    dev, dt = lay.device, lay.dtype
    H, S, T, D, P, Sl, me, Hl = lay.H, lay.S, lay.T, lay.D, lay.P, lay.Sl, lay.rank, lay.Hl
    order = list(head_order)
    if break_order and me == P - 1 and H > 1:
        order[0], order[-1] = order[-1], order[0]
    mine = order[lay.starts[me]:lay.starts[me + 1]]
    tok_local = torch.arange(me * Sl, (me + 1) * Sl)
    shards = [tag_rows(t, range(H), tok_local, 1, D, dt, dev) for t in range(3)]
    texts = [tag_rows(t, range(H), torch.arange(T), 2, D, dt, dev) for t in range(3)] if T else None
    out_shard = torch.zeros((H, Sl, D), dtype=dt, device=dev)
    out_text = torch.zeros((H, T, D), dtype=dt, device=dev) if T else None
    my_parts = None if parts is None else list(parts)[lay.starts[me]:lay.starts[me + 1]]

    sgs, _, _ = lay.grouping(len(groups))

    def attend(g0, g1, gi):  # identity attention: the output rows of a head slot are its query rows
        sg = sgs[gi]
        sg.buffer(bufs[3]).copy_(sg.buffer(bufs[0]))
        for i in range(g0, g1):  # a partial slot: only the query tokens of its range
            if my_parts is not None and my_parts[i] is not None:
                t0, t1 = my_parts[i]
                o_i = sg.head_view(bufs[3])[i - g0]
                keep = torch.zeros(S, dtype=torch.bool, device=dev)
                keep[t0:t1] = True
                o_i[sg.lay.row_map[:S].long()[~keep]] = 0
                if T and t1 != S:  # only the part that ends at the last video token answers the text queries
                    o_i[sg.lay.row_map[S:S + T].long()] = 0

    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    exchange_and_attend(lay, shards, bufs, order, texts, groups, attend, out_shard, out_text, vwire=vwire, parts=parts)
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) * 1e3

    failed = []
    all_tok = torch.arange(S)
    n16 = 2 if vwire is not None else 3
    for sg in sgs:
        rm = sg.lay.row_map.long()
        mine_g = mine[sg.g0:sg.g1]
        for t in range(n16):
            got = sg.head_view(bufs[t])
            if not torch.equal(got[:, rm[:S]], tag_rows(t, mine_g, all_tok, 1, D, dt, dev)):
                failed.append("qkv"[t] + " video rows")
            if T and not torch.equal(got[:, rm[S:S + T]], tag_rows(t, mine_g, torch.arange(T), 2, D, dt, dev)):
                failed.append("qkv"[t] + " text rows")
    if vwire is not None:  # the bytes and scales of ONE conversion over the assembled sequence of this rank's heads
        v_full = tag_rows(2, mine, all_tok, 1, D, dt, dev)
        amax = torch.zeros((Hl, D), dtype=torch.float32, device=dev)
        ops.fp8_v_absmax(v_full, amax)
        if T:
            v_txt = tag_rows(2, mine, torch.arange(T), 2, D, dt, dev)
            ops.fp8_v_absmax(v_txt, amax)
        exp8 = torch.empty((Hl, S, D), dtype=torch.uint8, device=dev)
        vd = torch.empty((Hl, D), dtype=torch.float32, device=dev)
        ops.fp8_v_convert(v_full, amax, exp8, v_descale=vd)
        if T:
            exp8t = torch.empty((Hl, T, D), dtype=torch.uint8, device=dev)
            ops.fp8_v_convert(v_txt, amax, exp8t)
        for sg in sgs:
            rm = sg.lay.row_map.long()
            got8 = sg.head_view(vwire.buf)
            if not torch.equal(got8[:, rm[:S]], exp8[sg.g0:sg.g1]):
                failed.append("v video rows (e4m3 on the wire)")
            if T and not torch.equal(got8[:, rm[S:S + T]], exp8t[sg.g0:sg.g1]):
                failed.append("v text rows (e4m3)")
            if not torch.equal(vwire.descale(sg), vd[sg.g0:sg.g1]):
                failed.append("v_descale")
    if not torch.equal(out_shard, shards[0]):
        failed.append("returned sequence shard")
    if T and not torch.equal(out_text, texts[0]):
        failed.append("gathered text rows")
    ok = torch.tensor([0 if failed else 1], dtype=torch.int32, device=dev)
    if (P > 1 or FORCE_COLLECTIVES) and not lay.loopback:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=lay.group)
    esz = torch.empty((), dtype=dt).element_size()
    away = (lay.Hv - Hl) * Sl * D  # elements of one tensor this rank sends to its peers
    back = Hl * (S - Sl) * D  # ... and receives from them
    nbytes = (n16 * esz + (1 if vwire is not None else 0)) * (away + back) + esz * (away + back)  # q, k, v in; o back
    nbytes += esz * (P - 1) * max(lay.counts) * T * D * 2 if T else 0  # text all-gather (padded to the largest count)
    return {"ok": bool(int(ok.item())), "failed_on_this_rank": failed, "bytes": int(nbytes), "ms": round(ms, 3),
            "what": f"tagged q,k,v through layer 0's exchange ({'v as e4m3, ' if vwire is not None else ''}"
                    f"{len(groups)} slot group(s), head counts {lay.counts}), identity attention, exact compare on every rank"}


_GROUP_STREAMS = {}


def _group_streams(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _GROUP_STREAMS:
        _GROUP_STREAMS[idx] = [torch.cuda.Stream(device=device)]
    return _GROUP_STREAMS[idx]


class VWire:
    """State of `v on the wire as e4m3` for one layout (fp8 attention under sequence parallelism): the sender converts
    its sequence shard of v with the scales of the WHOLE sequence -- per-(head, channel) abs-max of the shard
    (`ops.fp8_v_absmax`), one MAX all-reduce of H x D floats, `ops.fp8_v_convert` straight into destination head order
    -- so v crosses the links at half the bytes and lands as the kernels read it: 3.5 instead of 4 tensor-volumes per
    layer on the wire, and the receive-side quantiser touches q and k only.  The bytes and `v_descale` are the ones the
    receive-side conversion of the 16-bit sequence would produce (abs-max over shards = abs-max over the sequence)."""

    def __init__(self, lay: "UlyssesLayout", buf: torch.Tensor):
        dev = lay.device
        self.buf = buf  # (rows_total, D) uint8 receive buffer = the v8 operand of the attention kernels
        self.amax = torch.zeros((lay.H, lay.D), dtype=torch.float32, device=dev)
        # layouts of one slot count share this state, and with heads split by query range their total slot counts differ
        # from layer to layer: room for two extra parts per rank (`split_placement` max_parts), views sized by the layer's
        cap = lay.H + 2 * lay.P
        self._descale_all = torch.zeros((cap, lay.D), dtype=torch.float32, device=dev)
        self._stage = torch.empty((cap, lay.Sl, lay.D), dtype=torch.uint8, device=dev)
        self.lay = lay

    @property
    def descale_all(self) -> torch.Tensor:  # (slots of the current layout, D), in head_order
        return self._descale_all[:self.lay.Hv]

    @property
    def stage(self) -> torch.Tensor:
        return self._stage[:self.lay.Hv]

    def descale(self, sg: "SlotGroup") -> torch.Tensor:
        """v_descale rows of this rank's head slots of slot group `sg` (`descale_all` is in the send order of the exchange:
        group-major, rank by rank inside a group)"""
        b = sg.first + sg.lay.starts[sg.lay.rank]
        return self.descale_all[b:b + sg.lay.Hl]


class SlotGroup:
    """Slot group gi of a layout's G: a receive layout of its own -- `lay`: its sub-layout (this rank holds `lay.Hl` = g1 - g0
    of its slots; every rank's count of the group's slots in `lay.counts`), `row0`: where its rows start in the parent's
    buffers, `first`: where its slots start in the group-major send order."""
    __slots__ = ("lay", "row0", "first", "g0", "g1")

    def __init__(self, lay, row0, first, g0, g1):
        self.lay, self.row0, self.first, self.g0, self.g1 = lay, row0, first, g0, g1

    def buffer(self, buf: torch.Tensor) -> torch.Tensor:
        """this group's rows of a parent-layout buffer ((rows_total, D), or (1, rows_total[, D]) operand arrays)"""
        if buf.shape[0] == 1 and buf.dim() in (2, 3):
            return buf[:, self.row0:self.row0 + self.lay.rows_total]
        return buf[self.row0:self.row0 + self.lay.rows_total]

    def head_view(self, buf: torch.Tensor) -> torch.Tensor:
        return self.lay.head_view(self.buffer(buf))

    def text_view(self, buf: torch.Tensor) -> torch.Tensor:
        """(slots, T, D) view of the text rows parked behind the group's video rows"""
        l = self.lay
        b = self.buffer(buf)
        return b.as_strided((l.Hl, l.T, l.D), (l.Sl * l.D, l.D, 1), b.storage_offset() + l.rows_video * l.D)


def make_row_map(S: int, T: int, P: int, Hl: int, device) -> torch.Tensor:
    """row_map[s] for video tokens, then the T text tokens parked behind the video rows (head stride Sl)."""
    Sl = S // P
    s = torch.arange(S, dtype=torch.int64)
    video = (s // Sl) * (Hl * Sl) + s % Sl
    text = P * Hl * Sl + torch.arange(T, dtype=torch.int64)
    return torch.cat([video, text]).to(torch.int32).to(device)


class UlyssesLayout:
    """Buffer geometry shared by Q, K, V and O of one layer."""

    _ROW_MAPS: dict = {}

    def __init__(self, H: int, S: int, T: int, D: int, P: int, rank: int, device, dtype, group=None,
                 counts: Optional[Sequence[int]] = None, extra_slots: Optional[int] = None):
        """`counts` = heads per rank (default: H / P everywhere).  Rank j owns the heads order[starts[j]:starts[j + 1]] of
        the head order given to the exchange; this rank's receive layout has Hl = counts[rank] head slots.
        `extra_slots`: the number of extra parts of heads split by query range (`split_placement`: len(order) - H); when
        given, sum(counts) must be exactly H + extra_slots -- a mistaken counts vector is refused here, not deep inside the
        first gather.  None = not checked (sum(counts) >= H)."""
        if S % P:
            raise ValueError(f"sequence {S} must be divisible by the sequence-parallel size {P}")
        if counts is None:
            if H % P:
                raise ValueError(f"heads {H} and sequence {S} must be divisible by the sequence-parallel size {P}")
            counts = [H // P] * P
        counts = [int(c) for c in counts]
        if extra_slots is not None and sum(counts) != H + int(extra_slots):
            raise ValueError(f"head counts {counts} hold {sum(counts)} slots, the placement has {H} heads + {extra_slots} extra parts")
        if len(counts) != P or sum(counts) < H or min(counts) < 1:
            raise ValueError(f"head counts {counts} do not place {H} heads on {P} ranks (at least one each; more slots than "
                             "heads = heads split by query range, `split_placement`)")
        self.H, self.S, self.T, self.D, self.P, self.rank = H, S, T, D, P, rank
        self.Hv = sum(counts)  # head slots over all ranks: H, + one per extra part of a head split by query range
        self.counts = counts
        self.starts = [sum(counts[:j]) for j in range(P + 1)]
        self.even = all(c == counts[0] for c in counts)
        self.Hl, self.Sl = counts[rank], S // P
        if T > self.Sl:
            raise ValueError("text length must not exceed the per-rank sequence shard")
        self.rows_video = P * self.Hl * self.Sl
        self.rows_total = self.rows_video + self.Hl * self.Sl  # text region keeps the head stride Sl
        self.device, self.dtype, self.group = torch.device(device), dtype, group
        key = (S, T, P, self.Hl, str(self.device))  # one table per slot count (the geometry caches key on its address)
        if key not in UlyssesLayout._ROW_MAPS:
            UlyssesLayout._ROW_MAPS[key] = make_row_map(S, T, P, self.Hl, device)
        self.row_map = UlyssesLayout._ROW_MAPS[key]
        self._loopback = False
        self._parent = None  # the layout whose slot group this one is (`grouping`)
        self._groupings = {}

    # loopback: every pass of the exchange that runs on THIS rank's GPU (staging gathers, own-chunk copies, text rows, the
    # attention on the received layout, the un-permute) with the transfers themselves left out -- one rank's compute at
    # P > 1 on a single GPU (bench.py --emulate-rank); remote chunks keep what the buffers held.  A slot group follows its parent.
    @property
    def loopback(self) -> bool:
        return self._loopback if self._parent is None else self._parent.loopback

    @loopback.setter
    def loopback(self, value: bool):
        self._loopback = bool(value)

    def grouping(self, n_groups: int):
        """(slot groups, perm, inv) of this layout cut into `n_groups` slot groups (`slot_groups` of every rank's head count).
        Slot group gi is a receive layout of its own: rank j holds `slot_groups(counts[j], G)[gi]` of its slots, the group's
        rows follow those of the groups before it in the buffers ((P + 1) * Sl rows per slot: P video chunks + the text
        segment), its tokens are read through ITS row map (the chunk stride is the group's slot count, not Hl).  The send
        side orders the heads group-major -- group by group, rank by rank inside a group: `perm[x]` = the position in the
        head order (rank-major) of send position x, `inv` its inverse -- so a group's heads for all ranks are one
        contiguous block with the ranks' chunks in rank order: one all_to_all_single per tensor and group.
        One group: the layout itself, identity permutation."""
        G = max(1, int(n_groups))
        if G not in self._groupings:
            if G > min(self.counts):
                raise ValueError(f"{G} slot groups, but a rank holds only {min(self.counts)} heads")
            if G == 1:
                ident = list(range(self.Hv))
                self._groupings[G] = ([SlotGroup(self, 0, 0, 0, self.Hl)], ident, ident)
            else:
                cuts = [slot_groups(c, G) for c in self.counts]  # per rank: [(p0, p1)] per group
                sgs, perm, first = [], [], 0
                for gi in range(G):
                    counts_g = [cuts[j][gi][1] - cuts[j][gi][0] for j in range(self.P)]
                    for j in range(self.P):
                        perm += list(range(self.starts[j] + cuts[j][gi][0], self.starts[j] + cuts[j][gi][1]))
                    sub = UlyssesLayout(sum(counts_g), self.S, self.T, self.D, self.P, self.rank, self.device, self.dtype,
                                        self.group, counts=counts_g)
                    sub._parent = self
                    g0, g1 = cuts[self.rank][gi]
                    sgs.append(SlotGroup(sub, (self.P + 1) * self.Sl * g0, first, g0, g1))
                    first += sum(counts_g)
                inv = [0] * len(perm)
                for x, i in enumerate(perm):
                    inv[i] = x
                self._groupings[G] = (sgs, perm, inv)
        return self._groupings[G]

    def new_buffer(self) -> torch.Tensor:
        # zeros: the rows between a head's T text rows and the next head's are never written, and the fp8 conversion
        # takes the abs-max of the whole buffer
        return torch.zeros((self.rows_total, self.D), dtype=self.dtype, device=self.device)

    def fp8_operands(self) -> "ops.Fp8Operands":
        """e4m3 operand buffers of the receive layout (allocated once per layout; `fp8_views(out=...)` fills them)"""
        nws = ops._C.lib().vorta_fp8_quant_ws_floats(self.Hl, self.D)
        return ops.Fp8Operands(*(torch.zeros((1, self.rows_total, self.D), dtype=torch.uint8, device=self.device) for _ in range(3)),
                               torch.zeros((self.Hl, self.D), dtype=torch.float32, device=self.device),
                               torch.zeros(nws, dtype=torch.float32, device=self.device))

    def fp8_views(self, bufs: Sequence[torch.Tensor], scale: Optional[float] = None, out=None,
                  vwire: Optional["VWire"] = None, v_descale: Optional[torch.Tensor] = None):
        """e4m3 copies of the q, k, v receive buffers for the fp8 attention kernels: ONE conversion of each whole buffer
        (the head views overlap, so converting per view would redo it Hl times) in the quantiser's segmented row layout
        -- row r belongs to head slot (r // Sl) % Hl, text rows behind the video rows -- so every local head keeps its
        own scales and key centre.  A slot group converts the rows of ITS layout when it has landed (scales are per head, so
        the bytes are the ones one call over all heads writes).  `vwire`: v arrived as e4m3 (`out.v` is its receive
        buffer, `v_descale` its scales): q and k only.  Returns (q8, k8, v8 head views, v_descale (Hl, D), operands)."""
        from ..routed import FP8_CENTER_K
        x = [b.view(1, self.rows_total, self.D) for b in bufs[:3]]
        if vwire is not None:
            if out is None or v_descale is None:
                raise ValueError("fp8_views(vwire=...): `out` = the operands whose v is the e4m3 receive buffer, `v_descale` its scales")
            x[2] = None
        f8 = ops.fp8_quantize_qkv(*x, scale, out=out, center_k=FP8_CENTER_K, heads=self.Hl, seg_len=self.Sl,
                                  tail_first=self.rows_video, tail_len=self.T)
        shape, stride = (self.Hl, self.rows_total - (self.Hl - 1) * self.Sl, self.D), (self.Sl * self.D, self.D, 1)
        hv = lambda t: t[0].as_strided(shape, stride, t[0].storage_offset())
        vd = f8.v_descale if vwire is None else v_descale
        return hv(f8.q), hv(f8.k), hv(f8.v), vd, f8

    def i8_operands(self) -> "ops.I8Operands":
        """int8 key buffers of the receive layout for precision "i8pv" (allocated once per layout; `i8_views` fills them)"""
        dev = self.device
        return ops.I8Operands(torch.zeros((1, self.rows_total, self.D), dtype=torch.int8, device=dev),
                              torch.zeros((1, self.rows_total), dtype=torch.float32, device=dev),
                              torch.zeros((self.Hl, 2, self.D), dtype=torch.float32, device=dev),
                              torch.ones((self.Hl,), dtype=torch.float32, device=dev),
                              torch.zeros(2 * self.Hl * self.D + self.Hl, dtype=torch.float32, device=dev))

    def i8_views(self, bufs: Sequence[torch.Tensor], out: "ops.I8Operands"):
        """int8 keys of the k receive buffer (q's is sampled for the statistics): ONE conversion of the buffer in the
        quantiser's segmented row layout, every local head with its own centres, balance vector and scale (a slot group:
        the rows of its layout, when it has landed).  Returns the I8Operands whose k8 / k_bias are head views."""
        ops.i8_quantize_k(bufs[0].view(1, self.rows_total, self.D), bufs[1].view(1, self.rows_total, self.D), out=out,
                          heads=self.Hl, seg_len=self.Sl, tail_first=self.rows_video, tail_len=self.T)
        rows = self.rows_total - (self.Hl - 1) * self.Sl
        k8 = out.k8[0].as_strided((self.Hl, rows, self.D), (self.Sl * self.D, self.D, 1), out.k8[0].storage_offset())
        kb = out.k_bias[0].as_strided((self.Hl, rows), (self.Sl, 1), out.k_bias[0].storage_offset())
        return ops.I8Operands(k8, kb, out.q_prep, out.k_head_scale, out.ws)

    def group_operands(self, sg: "SlotGroup", parent, cache: dict):
        """the 8-bit operand buffers (`fp8_operands` / `i8_operands` of this layout) as slot group `sg` sees them: its rows
        of the byte arrays, its heads of the per-head tensors, and a workspace of its own (the quantiser's workspace is
        laid out by head count, and groups convert on different streams).  `cache`: where the caller keeps them."""
        if sg.lay is self:
            return parent
        key = (id(parent), sg.row0, sg.lay.Hl)
        got = cache.get(key)
        if got is None:
            n, D, dev = sg.lay.Hl, self.D, self.device
            if isinstance(parent, ops.I8Operands):
                got = ops.I8Operands(sg.buffer(parent.k8), sg.buffer(parent.k_bias), parent.q_prep[sg.g0:sg.g1],
                                     parent.k_head_scale[sg.g0:sg.g1], torch.zeros(2 * n * D + n, dtype=torch.float32, device=dev))
            else:
                nws = ops._C.lib().vorta_fp8_quant_ws_floats(n, D)
                got = ops.Fp8Operands(sg.buffer(parent.q), sg.buffer(parent.k), sg.buffer(parent.v),
                                      parent.v_descale[sg.g0:sg.g1], torch.zeros(nws, dtype=torch.float32, device=dev))
            got = cache[key] = (got, parent)  # (the parent is kept alive: the key holds its id)
        return got[0]

    def head_view(self, buf: torch.Tensor) -> torch.Tensor:
        """(Hl, rows, D) overlapping view: head slot i starts i*Sl rows into the buffer."""
        return buf.as_strided((self.Hl, self.rows_total - (self.Hl - 1) * self.Sl, self.D),
                              (self.Sl * self.D, self.D, 1), buf.storage_offset())

    def _peer(self, j: int) -> int:
        return dist.get_global_rank(self.group, j) if self.group is not None else j

    def _staged(self) -> bool:
        # rehearsal transport: gloo cannot send/recv device memory, so stage through the host.  Only used when
        # the process group is gloo but the tensors live on a GPU (tests / 1-GPU rehearsals); RCCL is direct.
        return (not self.loopback) and self.device.type == "cuda" and dist.get_backend(self.group) == "gloo"

    @staticmethod
    def _finish(handle):
        """Make the current stream wait for collectives started by `_start_a2a` / `_start_allreduce_max` (no host
        synchronisation under RCCL)."""
        if handle is not None and handle[0] == "event":  # (the emulated wire of a loopback rank)
            torch.cuda.current_stream().wait_event(handle[1])
        elif handle is not None:
            for r in handle[1]:
                r.wait()

    def _start_a2a(self, pairs, in_rows: Optional[Sequence[int]] = None, out_rows: Optional[Sequence[int]] = None):
        """One `all_to_all_single` per (input, output) pair -- the collective the reference uses
        (vorta/ulysses/utils.py:47,79) -- on row blocks whose P chunks are contiguous on both sides (chunk j of the
        input goes to rank j, chunk j of the output comes from rank j; the own chunk is copied by the collective).
        `in_rows` / `out_rows`: rows per chunk when the ranks hold different numbers of heads (default: equal chunks).
        Asynchronous under RCCL; returns a handle for `_finish`."""
        me = self.rank
        if self.P == 1 and not FORCE_COLLECTIVES and not self.loopback:  # a world of one: the collective is a copy
            for i, o in pairs:
                o.copy_(i)
            return None
        if self.loopback:  # the own chunk is what the collective would have copied locally
            for i, o in pairs:
                ni = list(in_rows) if in_rows is not None else [i.shape[0] // self.P] * self.P
                no = list(out_rows) if out_rows is not None else [o.shape[0] // self.P] * self.P
                i0, o0 = sum(ni[:me]), sum(no[:me])
                o[o0:o0 + no[me]].copy_(i[i0:i0 + ni[me]])
            if EMULATE_LINK_GBPS > 0 and self.device.type == "cuda" and self.P > 1:
                # the pairs of one call follow each other on every link; a pair's largest chunk to another rank sets its time
                per_link = 0.0
                for i, o in pairs:
                    rows = list(in_rows) if in_rows is not None else [i.shape[0] // self.P] * self.P
                    per_link += max(r for j, r in enumerate(rows) if j != me) * i.shape[1] * i.element_size()
                return _wire_sleep(self.device, per_link)
            return None
        kw = {} if in_rows is None else dict(input_split_sizes=list(in_rows), output_split_sizes=list(out_rows))
        if not self._staged():
            return ("works", [dist.all_to_all_single(o, i, group=self.group, async_op=True, **kw) for i, o in pairs])
        for i, o in pairs:
            hi = i.detach().to("cpu")
            ho = torch.empty(o.shape, dtype=o.dtype)
            dist.all_to_all_single(ho, hi, group=self.group, **kw)
            o.copy_(ho)
        return None

    def _start_allreduce_max(self, x: torch.Tensor):
        """MAX all-reduce of a small float tensor over the sequence-parallel group; handle for `_finish`"""
        if self.loopback or (self.P == 1 and not FORCE_COLLECTIVES):
            return None
        if not self._staged():
            return ("works", [dist.all_reduce(x, op=dist.ReduceOp.MAX, group=self.group, async_op=True)])
        h = x.detach().to("cpu")
        dist.all_reduce(h, op=dist.ReduceOp.MAX, group=self.group)
        x.copy_(h)
        return None

    def _head_map(self, heads: Sequence[int]) -> torch.Tensor:
        """int32 device copy of a head list for `ops.permute_heads`, built once per distinct list (a host-to-device copy
        per layer would stall the stream)."""
        maps = self.__dict__.setdefault("_head_maps", {})
        key = tuple(heads)
        m = maps.get(key)
        if m is None:
            if len(maps) > 4096:
                maps.clear()
            m = maps[key] = torch.tensor(key, dtype=torch.int32, device=self.device)
        return m

    def _stage(self, key):
        """(H, Sl, D) staging buffers in head_order (one per tensor slot), allocated once per layout."""
        st = self.__dict__.setdefault("_stages", {})
        if key not in st:
            st[key] = torch.empty((self.Hv, self.Sl, self.D), dtype=self.dtype, device=self.device)
        return st[key]

    @staticmethod
    def _run_of(order: Sequence[int], lo: int, n: int) -> Optional[int]:
        """first head if order[lo:lo+n] is a run of consecutive heads, else None"""
        a = order[lo]
        return a if all(order[lo + i] == a + i for i in range(n)) else None

    # ---- sequence shards -> head shards -------------------------------------------------------------
    def scatter_heads(self, shards: Sequence[torch.Tensor], bufs: Sequence[torch.Tensor], head_order: Sequence[int],
                      texts: Optional[Sequence[torch.Tensor]] = None):
        """shards[t]: (H, Sl, D) sequence shard of tensor t (q, k, v; any strides); bufs[t]: its layout buffer.
        ONE collective per tensor: the Hl heads bound for rank j are contiguous on the wire and land as the contiguous
        row block [j*Hl*Sl, (j+1)*Hl*Sl) of the receiver's buffer (per-head messages would be 2(P-1)Hl operations: ~1000
        per layer at P=8, H=24, a host-side cost of the same order as the layer's compute).  The sender needs the heads
        in destination order: one gather pass (`vorta_permute_heads` into a staging buffer) orders all heads at once --
        the projection output is a strided (S, H*D) view anyway, so this replaces the per-head `.contiguous()` -- or none
        when a contiguous shard already is in that order."""
        self._finish(self.scatter_heads_start(shards, bufs, head_order, texts)[0])

    def scatter_heads_start(self, shards: Sequence[torch.Tensor], bufs: Sequence[torch.Tensor],
                            head_order: Sequence[int], texts: Optional[Sequence[torch.Tensor]] = None,
                            groups: Optional[Sequence[Sequence[int]]] = None, vwire: Optional[VWire] = None):
        """`scatter_heads` split by local head slots: `groups` = [(slot0, slot1), ...] = `slot_groups(Hl, G)` (default: one
        group with all Hl slots).  One all_to_all_single per tensor is started per slot group, in order, into the group's
        own receive layout (`grouping`); returns the groups' handles, so the attention over the slots of group g can be
        enqueued after `_finish(handles[g])` while later groups are still in flight.  `vwire`: the third tensor (v) is
        converted to e4m3 on this side and travels into `vwire.buf` (bufs[2] is not used): abs-max of the shard, MAX
        all-reduce (in flight under the q/k staging pass), conversion into send order (under group 0's q/k transfers)."""
        Hl, Sl, me, P = self.Hl, self.Sl, self.rank, self.P
        starts, counts = self.starts, self.counts
        groups = [(0, Hl)] if groups is None else [tuple(g) for g in groups]
        if groups != slot_groups(Hl, len(groups)):
            raise ValueError(f"slot groups {groups} are not slot_groups({Hl}, {len(groups)})")
        if len(groups) > min(counts):
            raise ValueError(f"{len(groups)} slot groups, but a rank holds only {min(counts)} heads")
        if len(head_order) != self.Hv:
            raise ValueError(f"head order of {len(head_order)} slots, the layout has {self.Hv}")
        sgs, perm, _ = self.grouping(len(groups))
        send_order = [head_order[i] for i in perm]  # group-major (= head_order with one group)
        red = None
        v_shard = v_text = None
        if vwire is not None:
            v_shard, v_text = shards[2], (texts[2] if texts is not None and self.T else None)
            shards, bufs = shards[:2], bufs[:2]
            texts = texts[:2] if texts is not None else None
            vwire.amax.zero_()
            ops.fp8_v_absmax(v_shard, vwire.amax)
            if v_text is not None:  # replicated: every rank adds the same rows
                ops.fp8_v_absmax(v_text, vwire.amax)
            red = self._start_allreduce_max(vwire.amax)
        srcs = []
        staged = []
        in_place = len(groups) == 1 and self.even and self.Hv == self.H and send_order == list(range(self.H))
        for t, (x, buf) in enumerate(zip(shards, bufs)):
            if in_place and x.is_contiguous():  # already in destination order: sent straight from the shard
                src = x
            else:
                src = self._stage(("s", t))
                staged.append((x, src))
            srcs.append((src, buf))
        if staged:  # one gather pass orders the heads of every staged tensor
            if staged[0][0].is_cuda and HIP_STAGING:
                ops.permute_heads([x for x, _ in staged], [y for _, y in staged], src_map=self._head_map(send_order))
            else:  # CPU rehearsal (gloo)
                idx = torch.as_tensor(send_order, device=staged[0][0].device)
                for x, y in staged:
                    torch.index_select(x, 0, idx, out=y)
        mine = head_order[starts[me]:starts[me + 1]]
        if texts is not None and self.T:  # texts[t]: (H, T, D) replicated; its rows follow each local head slot's video
            for sg in sgs:
                mine_g = mine[sg.g0:sg.g1]
                if texts[0].is_cuda and HIP_STAGING:
                    ops.permute_heads(list(texts), [sg.text_view(buf) for buf in bufs], src_map=self._head_map(mine_g))
                else:
                    for t, buf in zip(texts, bufs):
                        sg.text_view(buf).copy_(t[torch.as_tensor(list(mine_g), device=t.device)])

        def convert_v():
            self._finish(red)
            ops.fp8_v_convert(v_shard, vwire.amax, vwire.stage, src_map=self._head_map(send_order),
                              v_descale=vwire.descale_all)
            if v_text is not None:
                for sg in sgs:
                    ops.fp8_v_convert(v_text, vwire.amax, sg.text_view(vwire.buf), src_map=self._head_map(mine[sg.g0:sg.g1]))
            return (vwire.stage, vwire.buf)

        def start(sg, pairs):
            # chunk j of the send block = rank j's slots of this group; every peer sends this rank's slots of it
            l = sg.lay
            blk = l.Hl * Sl
            splits = (None, None) if l.even else ([c * Sl for c in l.counts], [blk] * P)
            return l._start_a2a([(src.view(self.Hv * Sl, self.D)[sg.first * Sl:(sg.first + l.Hv) * Sl],
                                  sg.buffer(buf)[:P * blk]) for src, buf in pairs], *splits)

        def join(h, hv):
            if hv is not None and hv[0] == "event":  # (emulated wire: the second sleep follows the first on the wire stream)
                return hv
            return None if h is None and hv is None else ("works", (h[1] if h else []) + (hv[1] if hv else []))

        handles = []
        v_pair = None
        for gi, sg in enumerate(sgs):
            h = start(sg, srcs)
            if vwire is not None:
                if v_pair is None:  # converted while group 0's q and k are on the links
                    v_pair = convert_v()
                h = join(h, start(sg, [v_pair]))
            handles.append(h)
        return handles

    # ---- head shards -> sequence shards -------------------------------------------------------------
    def gather_heads(self, buf: torch.Tensor, out_shard: torch.Tensor, head_order: Sequence[int],
                     out_text: Optional[torch.Tensor] = None):
        """inverse of scatter_heads for the attention output: out_shard (H, Sl, D), any strides.  One collective;
        received head blocks go straight into out_shard when it is contiguous and the heads are in natural order, else
        through a staging buffer and one un-permute pass."""
        state = self.gather_heads_begin(out_shard, head_order)
        self._finish(self.gather_heads_start(buf, state))
        self.gather_heads_end(buf, state, out_text)

    def gather_heads_begin(self, out_shard: torch.Tensor, head_order: Sequence[int],
                           parts: Optional[Sequence[Optional[tuple]]] = None, n_groups: int = 1):
        """`parts` (`split_placement`): parts[i] = (t0, t1) when slot i of the head order computed only those query tokens
        of its head; this rank's token shard then takes each row of the head from the slot whose range holds it.
        `n_groups`: the slot groups the output returns in (the staging buffer is in their group-major send order)."""
        if parts is not None and not any(x is not None for x in parts):
            parts = None
        if (parts is None) != (self.Hv == self.H):
            raise ValueError("a head order with split heads needs `parts`, and only such an order takes them")
        sgs, perm, inv = self.grouping(n_groups)
        send_order = [head_order[i] for i in perm]
        direct = (parts is None and len(sgs) == 1 and self.even and out_shard.is_contiguous()
                  and send_order == list(range(self.H)))
        dst = out_shard if direct else self._stage(("g", 0))
        state = dict(out_shard=out_shard, order=list(head_order), direct=direct, dst=dst, parts=None, sgs=sgs,
                     send_order=send_order)
        if parts is not None:
            lo, hi = self.rank * self.Sl, (self.rank + 1) * self.Sl  # this rank's tokens
            slot_of = [-1] * self.H
            extra = []  # (slot, head, first row, end row) of the shard: rows another part of the head computed
            best = {}
            for i, (h, pr) in enumerate(zip(head_order, parts)):
                t0, t1 = (0, self.S) if pr is None else pr
                r0, r1 = max(t0, lo) - lo, min(t1, hi) - lo
                if r1 <= r0:
                    continue
                if r1 - r0 > best.get(h, (0, 0))[0]:
                    if h in best:
                        extra.append(best[h][1])
                    best[h] = (r1 - r0, (i, h, r0, r1))
                    slot_of[h] = i
                else:
                    extra.append((i, h, r0, r1))
            if min(slot_of) < 0:
                raise ValueError("the parts of a split head do not cover this rank's tokens")
            # the text rows of a head come from the part that ends at the last video token (it owns the text queries)
            text_slot = [-1] * self.H
            for i, (h, pr) in enumerate(zip(head_order, parts)):
                if pr is None or pr[1] == self.S:
                    text_slot[h] = i
            if self.T and min(text_slot) < 0:
                raise ValueError("no part of a split head ends at the last video token (the owner of its text rows)")
            # slots are positions of the head order; the staging buffer holds them in send order (`inv`)
            state.update(parts=list(parts), slot_of=[inv[i] for i in slot_of],
                         extra=[(inv[i], h, r0, r1) for i, h, r0, r1 in extra], text_slot=text_slot)
        return state

    def gather_heads_start(self, buf: torch.Tensor, state, slots: Optional[Sequence[int]] = None, gi: int = 0,
                           n_groups: int = 1):
        """send the attention output of slot group `gi` (of the `n_groups` the state was begun with; `slots` = its local
        slot range, checked) back: one all_to_all_single; returns the handle"""
        Sl, P = self.Sl, self.P
        sgs = state["sgs"]
        if n_groups != len(sgs) or not 0 <= gi < len(sgs):
            raise ValueError(f"slot group {gi} of {n_groups}: the exchange was begun with {len(sgs)} groups")
        sg = sgs[gi]
        if slots is not None and tuple(slots) != (sg.g0, sg.g1):
            raise ValueError(f"slot group {gi} holds local slots {(sg.g0, sg.g1)}, not {tuple(slots)}")
        l = sg.lay
        blk = l.Hl * Sl
        splits = (None, None) if l.even else ([blk] * P, [c * Sl for c in l.counts])
        dst = state["dst"].view(self.Hv * Sl, self.D)[sg.first * Sl:(sg.first + l.Hv) * Sl]
        return l._start_a2a([(sg.buffer(buf)[:P * blk], dst)], *splits)

    def gather_heads_end(self, buf: torch.Tensor, state, out_text: Optional[torch.Tensor] = None):
        """after every slot group's handle was finished: un-permute (if staged) and all-gather the text rows"""
        Hl, Sl = self.Hl, self.Sl
        out_shard, head_order = state["out_shard"], state["order"]
        if state["parts"] is not None:
            # every head from the slot that computed most of this shard's rows, then the rows another part computed
            if out_shard.is_cuda and HIP_STAGING:
                ops.permute_heads([state["dst"]], [out_shard], src_map=self._head_map(state["slot_of"]))
            else:
                out_shard.copy_(state["dst"][torch.as_tensor(state["slot_of"], device=out_shard.device)])
            for i, h, r0, r1 in state["extra"]:
                out_shard[h, r0:r1].copy_(state["dst"][i, r0:r1])
        elif not state["direct"]:
            if out_shard.is_cuda and HIP_STAGING:
                ops.permute_heads([state["dst"]], [out_shard], dst_map=self._head_map(state["send_order"]))
            else:
                out_shard.index_copy_(0, torch.as_tensor(state["send_order"], device=out_shard.device), state["dst"])
        if out_text is not None and self.T:
            cap = max(self.counts)  # equal message sizes for the collective: padded to the largest head count
            local = torch.empty((cap, self.T, self.D), dtype=buf.dtype, device=buf.device)
            for sg in state["sgs"]:
                local[sg.g0:sg.g1].copy_(sg.text_view(buf))
            if cap > Hl:
                local[Hl:].zero_()
            parts = [torch.empty_like(local) for _ in range(self.P)]
            if self.loopback:
                parts = [local] * self.P
            elif (self.P > 1 or FORCE_COLLECTIVES) and self._staged():
                hp = [torch.empty(local.shape, dtype=local.dtype) for _ in range(self.P)]
                dist.all_gather(hp, local.cpu(), group=self.group)
                parts = [h.to(local.device) for h in hp]
            elif self.P > 1 or FORCE_COLLECTIVES:
                dist.all_gather(parts, local, group=self.group)
            else:
                parts = [local]
            if self.loopback:  # every peer's heads stand in with this rank's rows (timing only)
                allh = torch.cat([local[torch.arange(c, device=local.device) % Hl] for c in self.counts], dim=0)
            else:
                allh = torch.cat([parts[j][:self.counts[j]] for j in range(self.P)], dim=0)  # (H, T, D) in head_order
            if state["parts"] is not None:  # slots -> heads by the text owner of every head
                if out_text.is_cuda and HIP_STAGING:
                    ops.permute_heads([allh], [out_text], src_map=self._head_map(state["text_slot"]))
                else:
                    out_text.copy_(allh[torch.as_tensor(state["text_slot"], device=out_text.device)])
            elif out_text.is_cuda and HIP_STAGING:
                ops.permute_heads([allh], [out_text], dst_map=self._head_map(head_order))
            else:
                out_text[torch.as_tensor(list(head_order), device=out_text.device)] = allh


class _RankState:
    """everything of the receive side that depends on how many head slots this rank holds: buffers, e4m3 operands, the
    routed geometry composed with the layout's row map"""

    def __init__(self, lay: UlyssesLayout, cfg: dict, te: int, fp8: bool, v_wire: bool, loopback: bool):
        from ..routed import RoutedGeometry
        self.bufs = [lay.new_buffer() for _ in range(4)]  # q, k, v, o
        if loopback:  # the chunks no peer will fill: finite values of the same distribution
            for b in self.bufs[:3]:
                b.normal_()
        self.f8 = self.vwire = self.i8 = None
        if fp8 in ("fp8pv", "i8pv", "auto8"):  # only v is e4m3, converted on the send side (it always travels as bytes); "i8pv": k is
            # rounded to int8 on this side, slot group by slot group (q by the attention kernel)
            if not v_wire:
                raise ValueError(f"precision '{fp8}' under sequence parallelism converts v on the send side (v_wire)")
            if fp8 in ("i8pv", "auto8"):
                self.i8 = lay.i8_operands()
            buf8 = torch.zeros((lay.rows_total, lay.D), dtype=torch.uint8, device=lay.device)
            if loopback:
                buf8.random_(0, 120)
            self.vwire = VWire(lay, buf8)
        elif fp8:
            self.f8 = lay.fp8_operands()
            if v_wire:  # v travels as e4m3 straight into the operand buffer
                self.vwire = VWire(lay, self.f8.v[0])
                if loopback:
                    self.f8.v.random_(0, 120)  # finite e4m3 bytes in the chunks no peer fills
        self.cfg, self.te, self.lay = cfg, te, lay
        self._geoms, self.group_ops = {}, {}
        self.geom = self.geom_for(lay)

    def geom_for(self, lay: UlyssesLayout):
        """the routed geometry composed with a layout's row map (one per distinct slot count: slot groups of equal size share it)"""
        from ..routed import RoutedGeometry
        g = self._geoms.get(lay.Hl)
        if g is None:
            cfg = self.cfg
            g = self._geoms[lay.Hl] = RoutedGeometry(cfg["latent"], cfg["tile"], cfg["window"], cfg["group"], cfg["rate"],
                                                     lay.device, row_map=lay.row_map)
            g.prebuild(self.te if cfg["model"] == "hunyuan" else 0)
        return g


class UlyssesRoutedAttention:
    """bench.py's N>1 step: synthetic sequence shards -> scatter_heads -> routed attention on the local
    heads (zero-copy layout) -> gather_heads.  Also the template for the attention processors under SP.
    placement = "even": H/P heads on every rank (`balanced_head_order`); "uneven": `balanced_placement` -- the ranks'
    head counts follow the layer's routes, one layout (and one set of receive buffers) per distinct count."""

    def __init__(self, cfg: dict, layer_experts: Sequence[np.ndarray], cost_of_expert: dict, device, dtype,
                 rank: int, P: int, group=None, n_sets: int = 2, concurrent: bool = False, fused: bool = True,
                 sliding_block_rows: int = 0, groups: int = 1, loopback: bool = False, fp8: bool = False,
                 v_wire: bool = True, placement: str = "even", heaviest_rank: bool = False, kv_splits=1):
        """heaviest_rank (with loopback): every layer is run as the rank that carries the largest cost in THAT layer -- a
        P-GPU step waits for its slowest rank layer by layer, so this (not a fixed rank) is the compute side of it."""
        from ..routed import HeadRouting
        if placement not in ("even", "uneven", "split"):
            raise ValueError("placement is 'even', 'uneven' or 'split'")
        if heaviest_rank and not loopback:
            raise ValueError("heaviest_rank is an emulation mode (loopback)")
        self.fp8 = fp8
        H, T = cfg["heads"], cfg["text"]
        S = cfg["latent"][0] * cfg["latent"][1] * cfg["latent"][2]
        self.cfg, self.P, self.rank = cfg, P, rank
        self.concurrent, self.fused, self.sliding_block_rows = concurrent, fused, sliding_block_rows
        self.te = cfg["text_valid"]
        costs = [cost_of_expert["full"], cost_of_expert["lowres"], cost_of_expert["sliding"]]
        # kv_splits: 1 (default), a number, or "auto": per layer, from this rank's count of workgroups -- a rank with one or
        # two heads runs under one round of workgroups on 256 CUs, where key splits shorten the layer and query splits do not
        self.kv_splits_arg, self.kv_splits = kv_splits, []
        self.orders, self.routes, self.lays, self.groups, self.parts = [], [], [], [], []
        layouts, self.states = {}, {}
        self.max_over_mean = []  # per layer: heaviest rank's cost / mean cost (1.0 = perfectly balanced)
        for e in layer_experts:
            parts = None
            if placement == "even":
                order, counts = balanced_head_order(e, costs, P, groups), [H // P] * P
            elif placement == "uneven":
                order, counts = balanced_placement(e, costs, P, groups)
            else:
                # whole 256-row workgroups at full size; small rehearsal sequences need finer steps to move anything
                order, counts, parts = split_placement(e, costs, P, S, groups, align=split_align(S))
                if not any(x is not None for x in parts):
                    parts = None
            starts = [sum(counts[:j]) for j in range(P + 1)]
            loads = placement_loads(e, costs, order, counts, parts, S)
            self.max_over_mean.append(max(loads) * P / sum(loads))
            r = max(range(P), key=lambda j: (loads[j], -j)) if heaviest_rank else rank
            key = (tuple(counts), r)
            if key not in layouts:
                lay = UlyssesLayout(H, S, T, 128, P, r, device, dtype, group, counts=counts, extra_slots=len(order) - H)
                lay.loopback = loopback
                layouts[key] = lay
                if lay.Hl not in self.states:
                    self.states[lay.Hl] = _RankState(lay, cfg, self.te, fp8, v_wire, loopback)
            lay = layouts[key]
            sg = slot_groups(lay.Hl, min(groups, min(counts)))
            local = [int(e[h]) for h in order[lay.starts[r]:lay.starts[r + 1]]]
            local_parts = [None] * len(local) if parts is None else parts[lay.starts[r]:lay.starts[r + 1]]
            if kv_splits == "auto":
                s_low = (S // (cfg["group"][0] * cfg["group"][1] * cfg["group"][2])) * int(
                    cfg["group"][0] * cfg["group"][1] * cfg["group"][2] * (1 - cfg["rate"]))
                rows = {0: S + T, 1: s_low + T, 2: S}
                wgs = sum(-(-rows[x] // 256) for x in local)
                self.kv_splits.append(max(1, min(8, round(768 / max(wgs, 1)))) if wgs < 384 else 1)
            else:
                self.kv_splits.append(max(1, int(kv_splits)))
            self.orders.append(order)
            self.lays.append(lay)
            self.groups.append(sg)
            self.parts.append(parts)
            self.routes.append([HeadRouting.from_expert_ids(
                local[g0:g1], device, q_ranges={i - g0: local_parts[i] for i in range(g0, g1) if local_parts[i] is not None})
                for g0, g1 in sg])
        self.Sl = S // P
        self.sets = []
        for i in range(n_sets):
            gen = torch.Generator(device=device).manual_seed(1234 + 97 * i + rank)
            shards = [torch.randn((H, self.Sl, 128), generator=gen, device=device, dtype=dtype) for _ in range(3)]
            tg = torch.Generator(device=device).manual_seed(4321 + i)  # text is replicated: same on every rank
            texts = [torch.randn((H, T, 128), generator=tg, device=device, dtype=dtype) for _ in range(3)] if T else None
            self.sets.append((shards, texts))
        self.out_shard = torch.empty((H, self.Sl, 128), dtype=dtype, device=device)
        self.out_text = torch.empty((H, T, 128), dtype=dtype, device=device) if T else None

    def selfcheck(self, l: int = 0, break_order: bool = False) -> dict:
        """`exchange_selfcheck` on layer l's placement, buffers and transport (before the warm-up of bench.py's N > 1 run)"""
        lay = self.lays[l]
        st = self.states[lay.Hl]
        if st.vwire is not None:
            st.vwire.lay = lay
        return exchange_selfcheck(lay, self.orders[l], self.groups[l], st.bufs, vwire=st.vwire, break_order=break_order,
                                  parts=self.parts[l])

    def layer(self, l: int, exchange_only: bool = False):
        """one layer: exchange in, routed attention per slot group, exchange back.  `exchange_only`: every pass of the exchange
        and no attention (bench.py's breakdown of what the exchange costs by itself)"""
        from ..routed import routed_attention
        shards, texts = self.sets[l % len(self.sets)]
        lay = self.lays[l]
        st = self.states[lay.Hl]
        if st.vwire is not None:
            st.vwire.lay = lay  # the layouts of one slot count share the state; the head offsets are the layer's
        sgs, _, _ = lay.grouping(len(self.groups[l]))
        geoms = [st.geom_for(sg.lay) for sg in sgs]  # built (and their tables) before the slot groups fork onto their streams

        def attend(g0, g1, gi):
            if exchange_only:
                return
            sg = sgs[gi]
            sub = sg.lay
            b = [sg.buffer(x) for x in st.bufs]
            q, k, v, o = (sub.head_view(x) for x in b)
            views = None
            if self.fp8 in ("i8pv", "auto8"):  # k of the slot group that has landed -> int8; q as it landed; v arrived as e4m3
                i8 = sub.i8_views(b, lay.group_operands(sg, st.i8, st.group_ops))
                views = (q, i8.k8, sg.head_view(st.vwire.buf), st.vwire.descale(sg), i8)
                if self.fp8 == "auto8":  # + the 16-bit keys and the group's tail flags: each head to the kernel that holds it
                    views += (k, ops.i8_tail_flags(i8.k8, row_map=sub.row_map[:lay.S + lay.T]))
            elif self.fp8 == "fp8pv":  # 16-bit q, k as they landed; v arrived as e4m3
                views = (q, k, sg.head_view(st.vwire.buf), st.vwire.descale(sg))
            elif self.fp8:  # the slot group that has landed is converted while the next one is in flight
                f8 = lay.group_operands(sg, st.f8, st.group_ops)
                q8, k8, v8, vd, f8 = sub.fp8_views(b, out=f8, vwire=st.vwire,
                                                   v_descale=None if st.vwire is None else st.vwire.descale(sg))
                if sub is lay:
                    st.f8 = f8
                views = (q8, k8, v8, vd)
            routed_attention(q, k, v, self.routes[l][gi], geoms[gi], model=self.cfg["model"],
                             text_len=self.cfg["text"], text_valid=self.te, out=o, concurrent=self.concurrent,
                             fused=self.fused, sliding_block_rows=self.sliding_block_rows, fp8=False, fp8_views=views,
                             kv_splits=self.kv_splits[l])

        # every cached table of this layer exists before the slot groups fork onto their streams (the routing lists were
        # copied to the device in __init__)
        exchange_and_attend(lay, shards, st.bufs, self.orders[l], texts, self.groups[l], attend, self.out_shard,
                            self.out_text, vwire=st.vwire, parts=self.parts[l],
                            prepare=lambda: [g.prebuild(self.te if self.cfg["model"] == "hunyuan" else 0) for g in geoms])
