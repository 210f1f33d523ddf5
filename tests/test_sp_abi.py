"""CPU: libvorta_sp.so (include/vorta_sp.h) -- the Ulysses exchange on RCCL behind a C ABI.

No GPU here and no communicator: the library loads and exports every symbol the header declares, and its OPERATION LISTS
(`vorta_sp_plan_*`: exactly the sends and receives the collective issues) are executed by a numpy simulation of P ranks and
compared with the reference's all_to_all_4D maps -- the golden vectors G9 the reference itself produced under gloo
(tests/golden/g9_ulysses_maps.npz; /root/reference/vorta/ulysses/utils.py:15-93) and the oracle's restatement for P = 2, 4, 8
with B > 1."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import vorta_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_symbol_the_header_declares():
    from vorta_amd.ulysses import rccl_abi
    header = open(os.path.join(ROOT, "include", "vorta_sp.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(vorta_sp_[a-z0-9_]+)\s*\(", header))
    assert declared == set(rccl_abi.SYMBOLS), declared ^ set(rccl_abi.SYMBOLS)
    lib = rccl_abi.lib()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.vorta_sp_abi_version() == 1 and C.sizeof(rccl_abi.Op) == 24
    # argument errors never reach RCCL
    assert lib.vorta_sp_unique_id(None) == -1 and lib.vorta_sp_destroy(None) == -1
    assert lib.vorta_sp_init(None, 0, 1, None) == -1 and lib.vorta_sp_rank(None) == -1
    assert lib.vorta_sp_a2a_seq2head(None, None, None, 1, 8, 4, 128, 0, None) == -1
    assert lib.vorta_sp_plan_seq2head(0, 3, 1, 8, 4, 128, 2, None, 0) == -1  # H % P != 0
    assert lib.vorta_sp_plan_seq2head(2, 2, 1, 8, 4, 128, 2, None, 0) == -1  # rank >= P


def _simulate(plan, xs, out_shape):
    """run the ranks' operation lists: the k-th send of rank a to rank b lands in the k-th receive of rank b from rank a"""
    P = len(xs)
    eb = xs[0].dtype.itemsize
    sends = {}
    for a in range(P):
        raw = xs[a].reshape(-1).view(np.uint8)
        for peer, is_send, off, nbytes in plan(a):
            if is_send:
                sends.setdefault((a, peer), []).append(raw[off:off + nbytes])
    ys = []
    for b in range(P):
        y = np.full(int(np.prod(out_shape)) * eb, 0xEE, dtype=np.uint8)
        seen = {}
        for peer, is_send, off, nbytes in plan(b):
            if not is_send:
                k = seen.get(peer, 0)
                piece = sends[(peer, b)][k]
                assert piece.size == nbytes
                y[off:off + nbytes] = piece
                seen[peer] = k + 1
        for peer in range(P):  # every send was consumed
            assert seen.get(peer, 0) == len(sends.get((peer, b), []))
        ys.append(y.view(xs[0].dtype).reshape(out_shape))
    return ys


@pytest.mark.parametrize("P", [2, 4])
def test_operation_lists_reproduce_the_references_maps_golden(golden, P):
    from vorta_amd.ulysses import rccl_abi
    g = golden("g9_ulysses_maps")
    xs = [np.ascontiguousarray(g[f"P{P}_r{r}_x"]) for r in range(P)]
    B, H, Sl, D = xs[0].shape
    eb = xs[0].dtype.itemsize
    ys = _simulate(lambda r: rccl_abi.plan_seq2head(r, P, B, H, Sl, D, eb), xs, (B, H // P, P * Sl, D))
    for r in range(P):
        assert np.array_equal(ys[r], g[f"P{P}_r{r}_y"])  # what the reference's all_to_all_4D(1, 2) gave rank r
    zs = _simulate(lambda r: rccl_abi.plan_head2seq(r, P, B, H, Sl, D, eb), ys, (B, H, Sl, D))
    for r in range(P):
        assert np.array_equal(zs[r], g[f"P{P}_r{r}_z"]) and np.array_equal(zs[r], xs[r])


@pytest.mark.parametrize("P,B,H,Sl,D", [(2, 1, 6, 5, 4), (4, 2, 8, 3, 16), (8, 1, 24, 7, 8), (8, 2, 40, 3, 4), (1, 1, 3, 4, 8)])
def test_operation_lists_vs_oracle(P, B, H, Sl, D):
    from vorta_amd.ulysses import rccl_abi
    rng = np.random.default_rng(P * 100 + H)
    xs = [rng.standard_normal((B, H, Sl, D)).astype(np.float32) for _ in range(P)]
    ys = _simulate(lambda r: rccl_abi.plan_seq2head(r, P, B, H, Sl, D, 4), xs, (B, H // P, P * Sl, D))
    ref = O.ulysses_seq_to_head(xs)
    for r in range(P):
        assert np.array_equal(ys[r], ref[r])
    zs = _simulate(lambda r: rccl_abi.plan_head2seq(r, P, B, H, Sl, D, 4), ys, (B, H, Sl, D))
    back = O.ulysses_head_to_seq(ref)
    for r in range(P):
        assert np.array_equal(zs[r], back[r]) and np.array_equal(zs[r], xs[r])
    # one contiguous slice of Sl * D elements per (batch item, head, peer) and direction: no pack / unpack pass anywhere
    ops = rccl_abi.plan_seq2head(0, P, B, H, Sl, D, 4)
    assert len(ops) == 2 * P * B * (H // P) and all(o[3] == Sl * D * 4 for o in ops)
