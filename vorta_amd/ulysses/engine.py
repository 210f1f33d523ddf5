"""Zero-copy Ulysses sequence parallelism for the routed attention op (MI355X: RCCL send/recv over xGMI).

Reference flow per tensor (vorta/ulysses/utils.py:61-91): transpose+contiguous -> all_to_all_single ->
device sync -> transpose+contiguous, i.e. four extra HBM passes per tensor and a host stall, applied per
routed subset (hunyuan.py:153-155,425-427,483-485), which also forces h_e % P == 0.

Here every head of the local sequence shard is sent as ONE contiguous (S/P, D) message straight from the
projection output into its final place in the receiver's buffer, and the attention kernels read that
buffer in place through a row table (`row_map`): token s of local head slot i lives at row
    (s // Sl) * (Hl*Sl) + i*Sl + s % Sl                      (Sl = S/P, Hl = H/P)
so no pack/unpack kernel exists on either side.  The output travels back the same way.  All H heads are
resharded once, BEFORE routing (any expert mix works), and because messages are per head the head -> rank
placement is free: `balanced_head_order` gives every rank the same number of heads and a near-equal sum
of expert costs (routes depend only on the timestep, so they are known when the layer starts).
Text tokens are replicated: each rank copies its heads' text rows behind the video rows
(`shrink_dim` in the reference, hunyuan.py:158-160) and the text outputs are all-gathered over heads (:187).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist


def balanced_head_order(experts: Sequence[int], cost_of_expert: Sequence[float], P: int) -> List[int]:
    """Heads grouped by destination rank (rank j owns order[j*Hl:(j+1)*Hl]): exactly Hl heads per rank,
    greedy longest-processing-time on the expert costs, ascending heads inside a rank. Deterministic, so
    every rank computes the same order without communicating."""
    H = len(experts)
    assert H % P == 0, f"heads {H} must be divisible by the sequence-parallel size {P}"
    Hl = H // P
    load = [0.0] * P
    bins: List[List[int]] = [[] for _ in range(P)]
    for h in sorted(range(H), key=lambda i: (-cost_of_expert[int(experts[i])], i)):
        j = min((r for r in range(P) if len(bins[r]) < Hl), key=lambda r: (load[r], r))
        bins[j].append(h)
        load[j] += cost_of_expert[int(experts[h])]
    return [h for b in bins for h in sorted(b)]


def make_row_map(S: int, T: int, P: int, Hl: int, device) -> torch.Tensor:
    """row_map[s] for video tokens, then the T text tokens parked behind the video rows (head stride Sl)."""
    Sl = S // P
    s = torch.arange(S, dtype=torch.int64)
    video = (s // Sl) * (Hl * Sl) + s % Sl
    text = P * Hl * Sl + torch.arange(T, dtype=torch.int64)
    return torch.cat([video, text]).to(torch.int32).to(device)


class UlyssesLayout:
    """Buffer geometry shared by Q, K, V and O of one layer."""

    def __init__(self, H: int, S: int, T: int, D: int, P: int, rank: int, device, dtype, group=None):
        if H % P or S % P:
            raise ValueError(f"heads {H} and sequence {S} must be divisible by the sequence-parallel size {P}")
        self.H, self.S, self.T, self.D, self.P, self.rank = H, S, T, D, P, rank
        self.Hl, self.Sl = H // P, S // P
        if T > self.Sl:
            raise ValueError("text length must not exceed the per-rank sequence shard")
        self.rows_video = P * self.Hl * self.Sl
        self.rows_total = self.rows_video + self.Hl * self.Sl  # text region keeps the head stride Sl
        self.device, self.dtype, self.group = torch.device(device), dtype, group
        self.row_map = make_row_map(S, T, P, self.Hl, device)

    def new_buffer(self) -> torch.Tensor:
        return torch.empty((self.rows_total, self.D), dtype=self.dtype, device=self.device)

    def head_view(self, buf: torch.Tensor) -> torch.Tensor:
        """(Hl, rows, D) overlapping view: head slot i starts i*Sl rows into the buffer."""
        return buf.as_strided((self.Hl, self.rows_total - (self.Hl - 1) * self.Sl, self.D),
                              (self.Sl * self.D, self.D, 1))

    def _peer(self, j: int) -> int:
        return dist.get_global_rank(self.group, j) if self.group is not None else j

    def _staged(self) -> bool:
        # rehearsal transport: gloo cannot send/recv device memory, so stage through the host.  Only used when
        # the process group is gloo but the tensors live on a GPU (tests / 1-GPU rehearsals); RCCL is direct.
        return self.device.type == "cuda" and dist.get_backend(self.group) == "gloo"

    def _run(self, p2p):
        """p2p: list of ("send"|"recv", tensor, peer)."""
        if not p2p:
            return
        if not self._staged():
            ops = [dist.P2POp(dist.isend if k == "send" else dist.irecv, t, self._peer(j), self.group) for k, t, j in p2p]
            for r in dist.batch_isend_irecv(ops):
                r.wait()
            return
        host = [(k, t, t.detach().to("cpu") if k == "send" else torch.empty(t.shape, dtype=t.dtype), j) for k, t, j in p2p]
        ops = [dist.P2POp(dist.isend if k == "send" else dist.irecv, h, self._peer(j), self.group) for k, t, h, j in host]
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        for k, t, h, j in host:
            if k == "recv":
                t.copy_(h)

    # ---- sequence shards -> head shards -------------------------------------------------------------
    def scatter_heads(self, shards: Sequence[torch.Tensor], bufs: Sequence[torch.Tensor], head_order: Sequence[int],
                      texts: Optional[Sequence[torch.Tensor]] = None):
        """shards[t]: (H, Sl, D) sequence shard of tensor t (q, k, v); bufs[t]: its layout buffer.
        One grouped send/recv for all tensors of the layer."""
        Hl, Sl, me = self.Hl, self.Sl, self.rank
        p2p = []
        for x, buf in zip(shards, bufs):
            for j in range(self.P):
                for i in range(Hl):
                    src = x[head_order[j * Hl + i]]
                    if j == me:
                        buf[me * Hl * Sl + i * Sl: me * Hl * Sl + (i + 1) * Sl].copy_(src)
                    else:
                        p2p.append(("send", src if src.is_contiguous() else src.contiguous(), j))
            for j in range(self.P):
                if j != me:
                    for i in range(Hl):
                        p2p.append(("recv", buf[j * Hl * Sl + i * Sl: j * Hl * Sl + (i + 1) * Sl], j))
        if texts is not None and self.T:
            for t, buf in zip(texts, bufs):  # t: (H, T, D) replicated
                for i in range(Hl):
                    r0 = self.rows_video + i * Sl
                    buf[r0:r0 + self.T].copy_(t[head_order[me * Hl + i]])
        self._run(p2p)

    # ---- head shards -> sequence shards -------------------------------------------------------------
    def gather_heads(self, buf: torch.Tensor, out_shard: torch.Tensor, head_order: Sequence[int],
                     out_text: Optional[torch.Tensor] = None):
        """inverse of scatter_heads for the attention output: out_shard (H, Sl, D) contiguous per head."""
        Hl, Sl, me = self.Hl, self.Sl, self.rank
        p2p = []
        for j in range(self.P):
            for i in range(Hl):
                src = buf[j * Hl * Sl + i * Sl: j * Hl * Sl + (i + 1) * Sl]
                if j == me:
                    out_shard[head_order[me * Hl + i]].copy_(src)
                else:
                    p2p.append(("send", src, j))
        for j in range(self.P):
            if j != me:
                for i in range(Hl):
                    dst = out_shard[head_order[j * Hl + i]]
                    assert dst.is_contiguous()
                    p2p.append(("recv", dst, j))
        self._run(p2p)
        if out_text is not None and self.T:
            local = torch.stack([buf[self.rows_video + i * Sl: self.rows_video + i * Sl + self.T] for i in range(Hl)])
            parts = [torch.empty_like(local) for _ in range(self.P)]
            if self.P > 1 and self._staged():
                hp = [torch.empty(local.shape, dtype=local.dtype) for _ in range(self.P)]
                dist.all_gather(hp, local.cpu(), group=self.group)
                parts = [h.to(local.device) for h in hp]
            elif self.P > 1:
                dist.all_gather(parts, local, group=self.group)
            else:
                parts = [local]
            allh = torch.cat(parts, dim=0)  # (H, T, D) in head_order
            out_text[torch.as_tensor(list(head_order), device=out_text.device)] = allh


class UlyssesRoutedAttention:
    """bench.py's N>1 step: synthetic sequence shards -> scatter_heads -> routed attention on the local
    heads (zero-copy layout) -> gather_heads.  Also the template for the attention processors under SP."""

    def __init__(self, cfg: dict, layer_experts: Sequence[np.ndarray], cost_of_expert: dict, device, dtype,
                 rank: int, P: int, group=None, n_sets: int = 2, concurrent: bool = False, fused: bool = True):
        from ..routed import HeadRouting, RoutedGeometry
        H, T = cfg["heads"], cfg["text"]
        S = cfg["latent"][0] * cfg["latent"][1] * cfg["latent"][2]
        self.cfg, self.P, self.rank = cfg, P, rank
        self.concurrent, self.fused = concurrent, fused
        self.te = cfg["text_valid"]
        self.lay = UlyssesLayout(H, S, T, 128, P, rank, device, dtype, group)
        self.geom = RoutedGeometry(cfg["latent"], cfg["tile"], cfg["window"], cfg["group"], cfg["rate"],
                                   torch.device(device), row_map=self.lay.row_map)
        costs = [cost_of_expert["full"], cost_of_expert["lowres"], cost_of_expert["sliding"]]
        self.orders, self.routes = [], []
        for e in layer_experts:
            order = balanced_head_order(e, costs, P)
            self.orders.append(order)
            local = [int(e[h]) for h in order[rank * self.lay.Hl:(rank + 1) * self.lay.Hl]]
            self.routes.append(HeadRouting.from_expert_ids(local, device))
        self.sets = []
        for i in range(n_sets):
            gen = torch.Generator(device=device).manual_seed(1234 + 97 * i + rank)
            shards = [torch.randn((H, self.lay.Sl, 128), generator=gen, device=device, dtype=dtype) for _ in range(3)]
            tg = torch.Generator(device=device).manual_seed(4321 + i)  # text is replicated: same on every rank
            texts = [torch.randn((H, T, 128), generator=tg, device=device, dtype=dtype) for _ in range(3)] if T else None
            self.sets.append((shards, texts))
        self.bufs = [self.lay.new_buffer() for _ in range(4)]  # q, k, v, o
        self.out_shard = torch.empty((H, self.lay.Sl, 128), dtype=dtype, device=device)
        self.out_text = torch.empty((H, T, 128), dtype=dtype, device=device) if T else None
        if self.te or cfg["model"] == "wan":
            self.geom.sta_tables(self.te)

    def layer(self, l: int):
        from ..routed import routed_attention
        shards, texts = self.sets[l % len(self.sets)]
        order = self.orders[l]
        self.lay.scatter_heads(shards, self.bufs[:3], order, texts)
        q, k, v, o = (self.lay.head_view(b) for b in self.bufs)
        routed_attention(q, k, v, self.routes[l], self.geom, model=self.cfg["model"], text_len=self.cfg["text"],
                         text_valid=self.te, out=o, concurrent=self.concurrent, fused=self.fused)
        self.lay.gather_heads(self.bufs[3], self.out_shard, order, self.out_text)
