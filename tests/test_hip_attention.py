"""GPU parity: the HIP gather flash-attention kernel (through the C ABI) vs the CPU oracle and the golden
vectors.  Run with `-m gpu` on an MI355X."""
import math

import numpy as np
import pytest
import torch

from oracle import vorta_oracle as O

pytestmark = pytest.mark.gpu

from _util import ATOL_SAME, check, dev, pad128, rounded, to_dev  # noqa: E402


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("block_rows", [128, 256])
def test_dense_ragged(dtype, block_rows):
    from vorta_amd import ops
    rng = np.random.default_rng(1)
    H, Sq, Skv = 3, 333, 417
    q, k, v = rng.standard_normal((H, Sq, 128)), rng.standard_normal((H, Skv, 128)), rng.standard_normal((H, Skv, 128))
    n_kv, q_valid = 401, 300  # keys 401.. are padding; query rows 300.. must come out zero
    qd, kd, vd = to_dev(q, dtype), to_dev(k, dtype), to_dev(v, dtype)
    out = torch.full((H, Sq, 128), 7.0, dtype=dtype, device=dev())
    ops.attn_fwd(qd, kd, vd, out, n_q=Sq, n_kv=n_kv, q_valid=q_valid, block_rows=block_rows)
    torch.cuda.synchronize()
    ref = O.dense_attention(rounded(q, dtype), rounded(k, dtype), rounded(v, dtype), kv_valid=n_kv, q_valid=q_valid)
    check(out, ref, dtype)
    assert torch.all(out[:, q_valid:] == 0)


def test_dense_golden_hunyuan_and_wan(golden):
    from vorta_amd import ops
    g = golden("g6_dense_out")
    S = 384
    t, te = (int(x) for x in g["hy_text"])
    dtype = torch.bfloat16
    q, k, v = (to_dev(pad128(g[n][0]), dtype) for n in ("hy_q", "hy_k", "hy_v"))
    out = torch.empty_like(q)
    ops.attn_fwd(q, k, v, out, n_q=S + t, n_kv=S + te, q_valid=S + te, scale=1 / math.sqrt(16))
    ref = np.concatenate([g["hy_out"][0], g["hy_eout"][0]], axis=1)
    check(out[..., :16], ref, dtype, gold=True)
    assert torch.all(out[..., 16:] == 0) and torch.all(out[:, S + te:] == 0)
    # Wan self attention and cross attention (Sq != Skv)
    q, k, v = (to_dev(pad128(g[n][0]), dtype) for n in ("wan_q", "wan_k", "wan_v"))
    out = torch.empty_like(q)
    ops.attn_fwd(q, k, v, out, n_q=S, n_kv=S, scale=0.25)
    check(out[..., :16], g["wan_out"][0], dtype, gold=True)
    kc, vc = to_dev(pad128(g["wan_kc"][0]), dtype), to_dev(pad128(g["wan_vc"][0]), dtype)
    ops.attn_fwd(q, kc, vc, out, n_q=S, n_kv=kc.shape[1], scale=0.25)
    check(out[..., :16], g["wan_cross_out"][0], dtype, gold=True)


def test_head_list_offsets_and_strides():
    """head-slot indirection, device-side head count, row offsets, non-contiguous (B,S,H,D)-style strides."""
    from vorta_amd import ops
    dtype = torch.bfloat16
    rng = np.random.default_rng(2)
    H, S = 5, 200
    base = rng.standard_normal((3, S, H, 128))  # (qkv, S, H, D) storage, heads interleaved per token
    qkv = to_dev(base, dtype)
    q, k, v = (qkv[i].permute(1, 0, 2) for i in range(3))  # (H,S,D) views with stride_s = H*128
    out = torch.zeros((H, S, 128), dtype=dtype, device=dev())
    heads = torch.tensor([4, 1, 3], dtype=torch.int32, device=dev())
    count = torch.tensor([2], dtype=torch.int32, device=dev())  # only the first two slots are live
    ops.attn_fwd(q, k, v, out, head_list=heads, n_heads_dev=count, n_q=50, q_row_offset=120, n_kv=70,
                 kv_row_offset=30)
    torch.cuda.synchronize()
    r = [rounded(base[i].transpose(1, 0, 2), dtype) for i in range(3)]
    for h in (4, 1):
        ref = O.dense_attention(r[0][h, 120:170], r[1][h, 30:100], r[2][h, 30:100])
        check(out[h, 120:170], ref, dtype)
    assert torch.all(out[3] == 0) and torch.all(out[0] == 0) and torch.all(out[4, :120] == 0)


@pytest.mark.parametrize("n_splits", [3, 8])
def test_split_keys(n_splits):
    from vorta_amd import ops
    dtype = torch.bfloat16
    rng = np.random.default_rng(3)
    H, Sq, Skv = 2, 40, 1000
    q, k, v = rng.standard_normal((H, Sq, 128)), rng.standard_normal((H, Skv, 128)), rng.standard_normal((H, Skv, 128))
    qd, kd, vd = to_dev(q, dtype), to_dev(k, dtype), to_dev(v, dtype)
    out = torch.empty_like(qd)
    ops.attn_fwd(qd, kd, vd, out, n_q=Sq, n_kv=Skv - 7, q_valid=33, n_splits=n_splits)
    ref = O.dense_attention(rounded(q, dtype), rounded(k, dtype), rounded(v, dtype), kv_valid=Skv - 7, q_valid=33)
    check(out, ref, dtype)
    assert torch.all(out[:, 33:] == 0)


def test_row_tables_groups_and_duplicates():
    """q_rows / kv_rows indirection with query groups (each group its own key list) and duplicate rows."""
    from vorta_amd import ops
    dtype = torch.float16
    rng = np.random.default_rng(4)
    H, S = 2, 600
    q, k, v = (rng.standard_normal((H, S, 128)) for _ in range(3))
    n_groups, glen, n_kv = 3, 100, 150
    q_rows = rng.permutation(S)[: n_groups * glen].astype(np.int32)  # distinct rows, shared by heads
    kv_rows = np.stack([rng.permutation(S)[:n_kv] for _ in range(n_groups)]).astype(np.int32)
    rest = np.setdiff1d(np.arange(S), q_rows)
    n_dup_pos, n_dup = 40, 3
    dup = rest[: n_dup_pos * n_dup].reshape(n_dup_pos, n_dup).astype(np.int32)
    qd, kd, vd = to_dev(q, dtype), to_dev(k, dtype), to_dev(v, dtype)
    out = torch.zeros_like(qd)
    ops.attn_fwd(qd, kd, vd, out, n_q=n_groups * glen, q_group_len=glen, n_kv=n_kv,
                 q_rows=torch.as_tensor(q_rows, device=dev()), kv_rows=torch.as_tensor(kv_rows, device=dev()),
                 kv_rows_stride_g=n_kv, dup_rows=torch.as_tensor(dup, device=dev()), n_dup_pos=n_dup_pos)
    torch.cuda.synchronize()
    rq, rk, rv = rounded(q, dtype), rounded(k, dtype), rounded(v, dtype)
    ref = np.zeros((H, S, 128))
    for g in range(n_groups):
        rows = q_rows[g * glen:(g + 1) * glen]
        ref[:, rows] = O.dense_attention(rq[:, rows], rk[:, kv_rows[g]], rv[:, kv_rows[g]])
    for p in range(n_dup_pos):
        ref[:, dup[p]] = ref[:, q_rows[p]][:, None]
    check(out, ref, dtype)
    untouched = np.setdiff1d(rest, dup.reshape(-1))
    assert torch.all(out[:, torch.as_tensor(untouched, device=dev())] == 0)


def test_online_softmax_rescale_branch():
    """A late key that dwarfs every earlier score forces the running-max rescale (guide rule 26)."""
    from vorta_amd import ops
    dtype = torch.bfloat16
    rng = np.random.default_rng(5)
    H, Sq, Skv = 1, 64, 512
    q, k, v = rng.standard_normal((H, Sq, 128)), rng.standard_normal((H, Skv, 128)), rng.standard_normal((H, Skv, 128))
    for row, key in ((3, 200), (17, 450), (40, 70)):
        k[0, key] = q[0, row] * 4.0  # score ~ 4*|q|^2/sqrt(128) >> the rest, first seen in a late block
    qd, kd, vd = to_dev(q, dtype), to_dev(k, dtype), to_dev(v, dtype)
    out = torch.empty_like(qd)
    ops.attn_fwd(qd, kd, vd, out, n_q=Sq, n_kv=Skv)
    ref = O.dense_attention(rounded(q, dtype), rounded(k, dtype), rounded(v, dtype))
    check(out, ref, dtype)


@pytest.mark.parametrize("boost", [0.35, 0.8, 2.0])
def test_deferred_rescale_paths(boost):
    """The pipelined kernel keeps the old running max while a block raises it by <= 6 (log2 units) and rescales
    beyond: late keys that raise some rows' max by a little (no rescale, P > 1), by about the threshold, and by
    a lot (rescale) must all match the oracle."""
    from vorta_amd import ops
    dtype = torch.bfloat16
    rng = np.random.default_rng(6)
    H, Sq, Skv = 1, 96, 640
    q, k, v = rng.standard_normal((H, Sq, 128)), rng.standard_normal((H, Skv, 128)), rng.standard_normal((H, Skv, 128))
    for row, key in ((5, 300), (33, 500), (70, 620)):
        k[0, key] = q[0, row] * boost  # raw score ~ boost*|q|^2 -> scaled log2 growth ~ boost * 128 * 0.1275
    qd, kd, vd = to_dev(q, dtype), to_dev(k, dtype), to_dev(v, dtype)
    out = torch.empty_like(qd)
    ops.attn_fwd(qd, kd, vd, out, n_q=Sq, n_kv=Skv)
    ref = O.dense_attention(rounded(q, dtype), rounded(k, dtype), rounded(v, dtype))
    check(out, ref, dtype)


def test_bad_arguments_raise():
    from vorta_amd import ops
    q = torch.zeros((1, 64, 128), dtype=torch.bfloat16, device=dev())
    with pytest.raises(ValueError):
        ops.attn_fwd(q, q, q, q.clone(), n_q=64, n_kv=0)
    with pytest.raises(Exception):
        ops.attn_fwd(q[..., :64], q[..., :64], q[..., :64], q[..., :64].clone(), n_q=64, n_kv=64)  # head_dim 64
    with pytest.raises(ValueError):
        ops.attn_fwd(q, q, q, q.clone(), n_q=64, n_kv=64, block_rows=100)
    with pytest.raises(ValueError):
        ops.attn_fwd(q, q, q, q.clone(), n_q=64, n_kv=64, scale=0.0)


@pytest.mark.parametrize("S", [32760, 118800])
def test_full_size_properties(S):
    """Size-independent properties at BASELINE.json's full sequence lengths (one head, dense):
       * a value matrix that is constant along the sequence must be reproduced exactly (rows of P sum to 1);
       * the result does not depend on the order of the keys (kv_rows = a permutation)."""
    from vorta_amd import ops
    dtype = torch.bfloat16
    gen = torch.Generator(device="cpu").manual_seed(S)
    q = torch.randn((1, S, 128), generator=gen).to(dtype).to(dev())
    k = torch.randn((1, S, 128), generator=gen).to(dtype).to(dev())
    const = torch.randn((1, 1, 128), generator=gen).to(dtype).to(dev())
    out = torch.empty_like(q)
    ops.attn_fwd(q, k, const.expand(1, S, 128).contiguous(), out, n_q=S, n_kv=S)
    assert (out.float() - const.float()).abs().max().item() <= 2e-2
    v = torch.randn((1, S, 128), generator=gen).to(dtype).to(dev())
    ops.attn_fwd(q, k, v, out, n_q=S, n_kv=S)
    perm = torch.randperm(S, generator=gen).to(torch.int32).to(dev())
    out2 = torch.empty_like(q)
    ops.attn_fwd(q, k, v, out2, n_q=S, n_kv=S, kv_rows=perm)
    assert (out.float() - out2.float()).abs().max().item() <= 1e-2
    # spot check 64 rows against the oracle
    rows = torch.randint(0, S, (64,), generator=gen)
    ref = O.dense_attention(q[0, rows].double().cpu().numpy(), k[0].double().cpu().numpy(), v[0].double().cpu().numpy())
    got = out[0, rows.to(dev())].float().cpu().numpy()
    assert np.abs(got - ref).max() <= ATOL_SAME[dtype]


def test_wide_row_stride_takes_the_64bit_kernel():
    """K/V rows spread over more than the 2 GiB window of the pipelined kernel's 32-bit offsets: ops selects the
    plain kernel (64-bit addressing) and the C ABI refuses variant 2 for a contiguous range it cannot address."""
    from vorta_amd import _C, ops
    import ctypes as C
    dtype = torch.bfloat16
    Hh, n, stride = 1, 4096, 300_000  # 4096 rows x 600 kB = 2.4 GB per tensor
    torch.manual_seed(0)
    q = torch.randn((Hh, 256, 128), device=dev()).to(dtype)
    kc, vc = (torch.randn((Hh, n, 128), device=dev()).to(dtype) for _ in range(2))
    big = [torch.empty(n * stride + 128, dtype=dtype, device=dev()) for _ in range(2)]
    k, v = (b.as_strided((Hh, n, 128), (0, stride, 1)) for b in big)
    k.copy_(kc); v.copy_(vc)
    out, ref = torch.empty_like(q), torch.empty_like(q)
    a, _ = ops._attn_args(q, k, v, out, n_q=256, n_kv=n)
    assert a.variant == 1
    ops.attn_fwd(q, k, v, out, n_q=256, n_kv=n)
    ops.attn_fwd(q, kc, vc, ref, n_q=256, n_kv=n, variant=1)
    assert torch.equal(out, ref)
    a.variant = 2
    assert _C.lib().vorta_attn_fwd(C.byref(a), ops._stream()) == _C.VORTA_EUNSUPPORTED


@pytest.mark.parametrize("seed", range(64))
def test_randomised_launch_shapes(seed):
    """Random launch geometry -- head lists, query groups with their own key lists, row tables, duplicate lists,
    q_valid / n_kv tails, key splits, both workgroup sizes and both kernel bodies -- against the oracle."""
    from vorta_amd import ops
    rng = np.random.default_rng(1000 + seed)
    dtype = (torch.bfloat16, torch.float16)[seed % 2]
    H_buf = int(rng.integers(1, 5))
    S = int(rng.integers(40, 700))
    q, k, v = (rng.standard_normal((H_buf, S, 128)) for _ in range(3))
    heads = rng.permutation(H_buf)[: int(rng.integers(1, H_buf + 1))].astype(np.int32)
    use_groups = bool(rng.integers(0, 2))
    use_qtab = use_groups or bool(rng.integers(0, 2))
    use_kvtab = use_groups or bool(rng.integers(0, 2))
    if use_groups:
        n_groups = int(rng.integers(1, 5))
        glen = int(rng.integers(1, max(2, S // n_groups)))
        n_q = n_groups * glen
    else:
        n_groups, n_q = 1, int(rng.integers(1, S + 1))
        glen = n_q
    n_kv = int(rng.integers(1, S + 1))
    q_rows = rng.permutation(S)[:n_q].astype(np.int32) if use_qtab else None
    q_off = 0 if use_qtab else int(rng.integers(0, S - n_q + 1))
    kv_rows = np.stack([rng.permutation(S)[:n_kv] for _ in range(n_groups)]).astype(np.int32) if use_kvtab else None
    kv_arg = kv_rows if (kv_rows is None or use_groups) else kv_rows[0]  # 1-D = one list shared by all heads
    kv_off = 0 if use_kvtab else int(rng.integers(0, S - n_kv + 1))
    q_valid = int(rng.integers(0, n_q + 1)) if rng.integers(0, 3) == 0 else n_q
    n_splits = int(rng.integers(1, 4)) if (not use_groups and rng.integers(0, 3) == 0) else 1
    block_rows = int(rng.choice([0, 128, 256]))
    variant = int(rng.choice([1, 2]))
    written = q_rows if use_qtab else np.arange(q_off, q_off + n_q)
    rest = np.setdiff1d(np.arange(S), written)
    n_dup, n_dup_pos = 0, 0
    dup = None
    if use_qtab and n_splits == 1 and len(rest) >= 2 and rng.integers(0, 2):
        n_dup = int(rng.integers(1, 3))
        n_dup_pos = int(rng.integers(1, min(n_q, len(rest) // n_dup) + 1))
        dup = rng.permutation(rest)[: n_dup_pos * n_dup].reshape(n_dup_pos, n_dup).astype(np.int32)
    qd, kd, vd = to_dev(q, dtype), to_dev(k, dtype), to_dev(v, dtype)
    out = torch.full_like(qd, 7.0)
    t = lambda a: None if a is None else torch.as_tensor(a, device=dev())
    ops.attn_fwd(qd, kd, vd, out, n_q=n_q, q_group_len=glen if use_groups else 0, n_kv=n_kv, q_valid=q_valid,
                 head_list=t(heads), n_heads=len(heads), q_rows=t(q_rows), q_row_offset=q_off, kv_rows=t(kv_arg),
                 kv_row_offset=kv_off, kv_rows_stride_g=n_kv if use_groups else 0, dup_rows=t(dup), n_dup_pos=n_dup_pos,
                 n_splits=n_splits, block_rows=block_rows, variant=variant)
    torch.cuda.synchronize()
    rq, rk, rv = rounded(q, dtype), rounded(k, dtype), rounded(v, dtype)
    ref = np.full((H_buf, S, 128), 7.0)
    for h in heads:
        for g in range(n_groups):
            pos = np.arange(g * glen, min((g + 1) * glen, n_q))
            rows = written[pos]
            keys = kv_rows[g] if use_kvtab else np.arange(kv_off, kv_off + n_kv)
            o = O.dense_attention(rq[h:h + 1, rows], rk[h:h + 1, keys], rv[h:h + 1, keys])[0]
            o[pos >= q_valid] = 0.0  # rows past q_valid are written as zeros (hunyuan.py:176)
            ref[h, rows] = o
        for p in range(n_dup_pos):
            ref[h, dup[p]] = ref[h, written[p]]
    desc = dict(seed=seed, H_buf=H_buf, S=S, heads=heads.tolist(), n_q=n_q, glen=glen, n_kv=n_kv, q_valid=q_valid,
                n_splits=n_splits, block_rows=block_rows, variant=variant, qtab=use_qtab, kvtab=use_kvtab, n_dup=n_dup)
    got = out.float().cpu().numpy()
    assert np.abs(got - ref).max() <= ATOL_SAME[dtype], desc


def test_table_extents_are_checked_on_the_host():
    """the kernels trust their tables; ops refuses every shape that cannot cover the launch (a 2-D key table is one
    list PER HEAD SLOT -- a single shared list must be 1-D)"""
    from vorta_amd import ops
    q, k, v = (torch.randn((3, 200, 128), device=dev()).to(torch.bfloat16) for _ in range(3))
    out = torch.empty_like(q)
    i32 = lambda *shape: torch.zeros(shape, dtype=torch.int32, device=dev())
    for kw in (dict(kv_rows=i32(1, 100)),                       # 3 head slots, one row
               dict(kv_rows=i32(90)),                           # shorter than n_kv
               dict(q_rows=i32(2, 200)),                        # 2 rows for 3 head slots
               dict(q_rows=i32(150)),                           # shorter than n_q
               dict(q_group_len=50, kv_rows=i32(4, 100)),       # groups without a group stride
               dict(q_group_len=50, kv_rows=i32(3, 100), kv_rows_stride_g=100),  # 4 groups, 3 key lists
               dict(head_list=i32(2), n_heads=3),
               dict(q_rows=i32(200), dup_rows=i32(2, 10, 2))):  # per-slot duplicate lists for 2 of 3 slots
        with pytest.raises(ValueError):
            ops.attn_fwd(q, k, v, out, n_q=200, n_kv=100, **kw)


def test_integration_md_stub_runs():
    """the ctypes binding printed in INTEGRATION.md, executed as is (library path made absolute), gives the same
    bits as the package's own launcher, including zeroed rows past the valid length"""
    import os
    from vorta_amd import _C, ops
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = text[text.index("# vorta/attention/_hip.py"):text.index("# hunyuan.py:170-176 becomes")]
    code = code.replace('"libvorta_hip.so"', repr(_C.LIB_PATH))
    ns = {}
    exec(code, ns)
    torch.manual_seed(0)
    q, k, v = (torch.randn((1, 3, 300, 128), device=dev()).to(torch.bfloat16) for _ in range(3))
    got = ns["dense_attention"](q, k, v, 280)
    want = torch.empty_like(q)
    ops.attn_fwd(q[0], k[0], v[0], want[0], n_q=300, n_kv=280, q_valid=280)
    assert torch.equal(got, want) and torch.all(got[:, :, 280:] == 0)


@pytest.mark.parametrize("block_rows", [128, 256])
def test_query_groups_of_different_lengths(block_rows):
    """vorta_attn_args.q_block_table: one (group, first, end) row per workgroup; groups of 700, 40 and 300 positions, each
    with its own key list"""
    from vorta_amd import ops
    dtype = torch.bfloat16
    rng = np.random.default_rng(11)
    H, rows, n_kv = 2, 1400, 333
    x = [rng.standard_normal((H, rows, 128)) for _ in range(3)]
    q, k, v = (to_dev(a, dtype) for a in x)
    bounds = [(0, 700), (700, 740), (740, 1040)]
    q_rows = rng.permutation(rows)[:1040].astype(np.int32)
    kv_rows = np.stack([rng.permutation(rows)[:n_kv] for _ in bounds]).astype(np.int32)
    table = np.array([(g, p, min(p + block_rows, b)) for g, (a, b) in enumerate(bounds) for p in range(a, b, block_rows)],
                     dtype=np.int32)
    out = torch.zeros((H, rows, 128), dtype=dtype, device=dev())
    ops.attn_fwd(q, k, v, out, n_q=1040, n_kv=n_kv, q_rows=torch.as_tensor(q_rows, device=dev()),
                 kv_rows=torch.as_tensor(kv_rows, device=dev()), kv_rows_stride_g=n_kv, block_rows=block_rows,
                 q_block_table=torch.as_tensor(table, device=dev()), n_key_lists=3)
    torch.cuda.synchronize()
    r = [rounded(a, dtype) for a in x]
    ref = np.zeros((H, rows, 128))
    for g, (a, b) in enumerate(bounds):
        ref[:, q_rows[a:b]] = O.dense_attention(r[0][:, q_rows[a:b]], r[1][:, kv_rows[g]], r[2][:, kv_rows[g]])
    check(out, ref, dtype)
    with pytest.raises(ValueError):
        ops.attn_fwd(q, k, v, out, n_q=1040, n_kv=n_kv, q_block_table=torch.as_tensor(table, device=dev()), n_key_lists=3)
