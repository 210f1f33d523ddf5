"""CPU checks of the oracle's fp8 (e4m3) restatement (oracle/vorta_oracle.py: e4m3_*, fp8_quantize_qkv,
fp8_attn_launch).  The reference has no fp8 path: these pin the emulator to the golden-pinned oracle -- with the
probability rounding switched off it must BE softmax attention on the dequantised operands, launch semantics
(row tables, groups, duplicates, split keys) included."""
import numpy as np
import pytest

from oracle import vorta_oracle as O


def test_e4m3_grid_round_trip_and_rounding():
    b = np.arange(256, dtype=np.uint8)
    v = O.e4m3_decode(b)
    ok = ~np.isnan(v)
    assert ok.sum() == 254  # 0x7f and 0xff are the NaNs of e4m3fn
    assert (O.e4m3_encode(v[ok]) == b[ok]).all()
    assert np.nanmax(v) == 448.0 and v[1] == 2.0 ** -9 and v[0x38] == 1.0
    # values measured on the hardware converter (tools/probe_fp8.hip, v_cvt_pk_fp8_f32)
    x = np.array([449.0, 464.0, 0.00146484375, 0.0009765625, 272.0, 0.017, 1.0625, 1.1875, 63.9, 3.76])
    want = np.array([448.0, 448.0, 0.001953125, 0.0, 256.0, 0.017578125, 1.0, 1.25, 64.0, 3.75])
    assert (O.e4m3_round(x) == want).all()
    assert (O.e4m3_round(-x) == -want).all()
    # nearest: no grid point is closer than the chosen one
    rng = np.random.default_rng(0)
    y = rng.standard_normal(4096) * np.exp2(rng.integers(-12, 8, 4096))
    r = O.e4m3_round(y)
    grid = np.sort(v[ok])
    best = np.abs(np.clip(y, -448, 448)[:, None] - grid[None]).min(1)
    assert np.allclose(np.abs(np.clip(y, -448, 448) - r), best, rtol=0, atol=0)


def test_quantizer_folds_the_softmax_scale_and_balances_ranges():
    rng = np.random.default_rng(1)
    H, S, D = 3, 200, 128
    q = rng.standard_normal((H, S, D)) * np.array([0.5, 1.0, 7.0])[:, None, None]
    k = rng.standard_normal((H, S, D)) * np.array([3.0, 1.0, 0.2])[:, None, None]
    v = rng.standard_normal((H, S, D)) * np.linspace(0.1, 9.0, D)
    z = O.fp8_quantize_qkv(q, k, v)
    c0 = (1 / np.sqrt(D)) * 1.4426950408889634
    assert np.allclose(z["qmul"] * z["kmul"], c0, rtol=1e-6)
    q8, k8, v8 = (O.e4m3_decode(z[n]) for n in ("q8", "k8", "v8"))
    # both operand maxima sit at sqrt(c0 * amax_q * amax_k), up to one e4m3 rounding
    tgt = np.sqrt(c0 * np.abs(q).max((1, 2)) * np.abs(k).max((1, 2)))
    assert np.allclose(np.abs(q8).max((1, 2)), tgt, rtol=0.07) and np.allclose(np.abs(k8).max((1, 2)), tgt, rtol=0.07)
    assert np.abs(v8).max() == 240.0 and np.allclose(np.abs(v8).max(1), 240.0)
    # scores from the e4m3 operands are the exp2-domain scores of the originals within the format's precision
    s8 = np.einsum("hqd,hkd->hqk", q8, k8)
    s = np.einsum("hqd,hkd->hqk", q, k) * c0
    assert np.sqrt(((s8 - s) ** 2).mean()) / np.sqrt((s ** 2).mean()) < 0.06  # ~2.7 % rms per operand
    assert np.allclose(v8 * z["v_descale"][:, None, :], v, atol=np.abs(v).max() * 2.0 ** -4)
    zh = O.fp8_quantize_qkv(q, k, v, v_per_head=True)
    assert np.allclose(zh["v_descale"], zh["v_descale"][:, :1])


def test_key_centring_is_softmax_invariant_and_buys_back_the_common_component():
    """flags bit1 of vorta_fp8_quantize_qkv: subtracting one vector from every key of a head leaves the attention
    output alone; with a common component in the keys the e4m3 scores are far closer to the true ones after it"""
    rng = np.random.default_rng(3)
    H, S, D = 2, 300, 128
    q = rng.standard_normal((H, S, D))
    k = rng.standard_normal((H, S, D)) + 6.0 * rng.standard_normal((H, 1, D))
    v = rng.standard_normal((H, S, D))
    rows = O.fp8_center_rows(S, H)
    c = np.stack([k[h, rows[h]].mean(0) for h in range(H)]).astype(np.float32)
    assert np.allclose(O.dense_attention(q, k - c[:, None, :], v), O.dense_attention(q, k, v), atol=1e-9)
    c0 = (1 / np.sqrt(D)) * 1.4426950408889634
    s = np.einsum("hqd,hkd->hqk", q, k) * c0
    err = {}
    for name, kc in (("as is", None), ("centred", c)):
        z = O.fp8_quantize_qkv(q, k, v, k_center=kc)
        s8 = np.einsum("hqd,hkd->hqk", O.e4m3_decode(z["q8"]), O.e4m3_decode(z["k8"]))
        d = s8 - s
        d = d - d.mean(-1, keepdims=True)  # softmax sees the scores of a query up to a constant
        err[name] = np.sqrt((d ** 2).mean())
    assert err["centred"] < 0.3 * err["as is"], err
    # the rows that define the centre: evenly spaced, in the segmented layout only the head's own rows that hold data
    seg = O.fp8_center_rows(2 * 3 * 50 + 3 * 50, 3, seg_len=50, tail_first=300, tail_len=7)
    for h, r in enumerate(seg):
        assert len(r) and ((r // 50) % 3 == h).all() and ((r < 300) | (r % 50 < 7)).all()


def _operands(rng, rows, scale=1.0):
    # values on the e4m3 grid, as the kernels see them
    return O.e4m3_round(rng.standard_normal((rows, 128)) * scale)


@pytest.mark.parametrize("n_splits", [1, 3])
def test_launch_emulator_without_rounding_is_softmax_attention(n_splits):
    rng = np.random.default_rng(2)
    q, k, v = _operands(rng, 150, 0.6), _operands(rng, 333, 0.6), _operands(rng, 333, 3.0)
    vd = rng.uniform(0.01, 0.1, 128)
    out = np.full((150, 128), 9.0)
    O.fp8_attn_launch(q, k, v, out, vd, n_q=120, q_row_offset=20, n_kv=300, kv_row_offset=10, q_valid=100,
                      n_splits=n_splits, round_p=False)
    # the scores are already in the exp2 domain: softmax base 2 == natural softmax of z * ln 2
    ref = O._softmax_attend(q[20:120], k[10:310], v[10:310], scale=np.log(2.0)) * vd
    assert np.abs(out[20:120] - ref).max() < 1e-12
    assert (out[120:140] == 0).all() and (out[:20] == 9.0).all() and (out[140:] == 9.0).all()


def test_launch_emulator_tables_groups_and_duplicates():
    rng = np.random.default_rng(3)
    rows = 400
    q, k, v = _operands(rng, rows, 0.6), _operands(rng, rows, 0.6), _operands(rng, rows, 2.0)
    vd = np.ones(128)
    n_q, glen, n_kv = 96, 40, 130  # 3 groups (40, 40, 16), each with its own key list
    q_rows = rng.permutation(rows)[:n_q]
    kv_rows = np.stack([rng.permutation(rows)[:n_kv] for _ in range(3)])
    free = np.setdiff1d(np.arange(rows), q_rows)
    dup = rng.permutation(free)[:2 * 10].reshape(10, 2)  # the first 10 positions fan out to 2 more rows each
    out = np.zeros((rows, 128))
    O.fp8_attn_launch(q, k, v, out, vd, n_q=n_q, n_kv=n_kv, q_rows=q_rows, q_group_len=glen, kv_rows=kv_rows,
                      dup_rows=dup, n_dup_pos=10, round_p=False)
    for g in range(3):
        pos = np.arange(g * glen, min((g + 1) * glen, n_q))
        ref = O._softmax_attend(q[q_rows[pos]], k[kv_rows[g]], v[kv_rows[g]], scale=np.log(2.0))
        assert np.abs(out[q_rows[pos]] - ref).max() < 1e-12
    for p in range(10):
        assert (out[dup[p]] == out[q_rows[p]]).all()
    untouched = np.setdiff1d(free, dup.reshape(-1))
    assert (out[untouched] == 0).all()


def test_probability_rounding_is_bounded_and_reference_points_move():
    """the rounded path stays within the e4m3 precision of the exact one, P' never exceeds 2^(p_bias+defer), and a
    row whose scores grow by more than `defer` moves the reference of its whole wave"""
    rng = np.random.default_rng(4)
    q, k, v = _operands(rng, 64, 0.8), _operands(rng, 512, 0.8), _operands(rng, 512, 2.0)
    k[300:] = O.e4m3_round(k[300:] * 3.0)  # later blocks score higher: forces the deferred rescale
    vd = np.ones(128)
    a, b = np.zeros((64, 128)), np.zeros((64, 128))
    O.fp8_attn_launch(q, k, v, a, vd, n_q=64, n_kv=512, round_p=True)
    O.fp8_attn_launch(q, k, v, b, vd, n_q=64, n_kv=512, round_p=False)
    assert 0 < np.abs(a - b).max() < 0.25 * np.abs(b).max()
    Ow, lw, mw = O._fp8_flash_rows(q[:32], k, v, 0, 8, 5.0, 3.0, True)
    z = q[:32] @ k.T
    assert (mw <= z.max(1) + 1e-12).all() and (mw >= z.max(1) - 3.0 - 1e-12).all()
    assert (mw > (q[:32] @ k[:64].T).max(1)).any()


def test_launch_emulator_groups_of_different_lengths():
    rng = np.random.default_rng(5)
    rows = 300
    q, k, v = _operands(rng, rows, 0.6), _operands(rng, rows, 0.6), _operands(rng, rows, 2.0)
    bounds = [(0, 70), (70, 100), (100, 230)]
    kv_rows = np.stack([rng.permutation(rows)[:90] for _ in bounds])
    q_rows = rng.permutation(rows)[:230]
    out = np.zeros((rows, 128))
    O.fp8_attn_launch(q, k, v, out, np.ones(128), n_q=230, n_kv=90, q_rows=q_rows, kv_rows=kv_rows, q_group_bounds=bounds,
                      round_p=False)
    for g, (a, b) in enumerate(bounds):
        ref = O._softmax_attend(q[q_rows[a:b]], k[kv_rows[g]], v[kv_rows[g]], scale=np.log(2.0))
        assert np.abs(out[q_rows[a:b]] - ref).max() < 1e-12
