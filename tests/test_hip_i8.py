"""GPU parity of the INT8-SCORE attention (csrc/attn_fwd_i8.hip + csrc/i8_quant.hip; include/vorta_hip.h ABI 6-7): scores on
v_mfma_i32_32x32x32_i8 with one scale per key row and per query row, P V in e4m3.  Gates:
  (q)   the quantiser against the oracle's float32 restatement: int8 bytes, row biases, query centre / balance vector, head
        scale and key centre BIT FOR BIT (plain layout); the segmented Ulysses layout and slot groups write the bytes of the
        plain call;
  (i)   kernel vs the oracle's emulator on the SAME operands -- per wave u (q8 . k8 + seed) with the wave's own query
        conversion restated (O.i8_wave_operands), v decoded from the e4m3 bytes -- with the probabilities' bytes written as the
        kernel writes them (rint(8 log2 P' + 56 - 8 e), one block exponent e per lane) at the kernel's reference points: the 16-bit tolerances plus the emulator's
        midpoint slack (dense ragged, tables / groups / duplicates / head lists, the rescale branch, split keys);
  (i')  kernel vs exact attention on the same operands: rel. Frobenius <= 4e-2;
  (ii)  operator PSNR against the bf16 kernels on every input family of tests/_fp8_inputs.py, both geometries, every expert:
        >= 40 dB over max|x| (>= 39 on the outlier-weights-with-common-part family) -- the targets VERDICT r03 item 2 set, not
        the measurements."""
import math

import numpy as np
import pytest
import torch

from oracle import vorta_oracle as O

pytestmark = pytest.mark.gpu

from _util import dev, rel_fro, to_dev  # noqa: E402
from test_hip_fp8 import RELF_PACK, _check, _vmax  # noqa: E402


RELF_DIRECT = 4e-2  # kernel vs exact-P attention on the same operands: e4m3 rounding of P (2.0-2.4e-2) + the linear mantissa


def _hooks(qd, i8, scale=None):
    """per head h: the `wave_operands` hook of O.fp8_attn_launch for that head's operands (numpy copies made once)"""
    q_np = qd.float().cpu().numpy()
    k8, kb = i8.k8.cpu().numpy(), i8.k_bias.cpu().numpy()
    qp, sk = i8.q_prep.cpu().numpy(), i8.k_head_scale.cpu().numpy()
    return [(lambda qr, kr, h=h: O.i8_wave_operands(q_np[h][qr], qp[h], sk[h], k8[h][kr], kb[h][kr], scale)) for h in range(qd.shape[0])]


def _vdec(v8, vd):
    return O.e4m3_decode(v8.cpu().numpy()), vd.cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_i8_quantizer_bit_for_bit_and_layouts(dtype):
    from vorta_amd import ops
    rng = np.random.default_rng(5)
    H, S = 3, 2600
    q = rng.standard_normal((H, S, 128)) * np.linspace(0.2, 4.0, 128) + 1.5 * rng.standard_normal((H, 1, 128))
    k = rng.standard_normal((H, S, 128)) * np.linspace(3.0, 0.3, 128) + 2.0 * rng.standard_normal((H, 1, 128))
    k[1, 17] = 0.0
    qd, kd = to_dev(q, dtype), to_dev(k, dtype)
    got = ops.i8_quantize_k(qd, kd)
    torch.cuda.synchronize()
    ref = O.i8_quantize_k(qd.float().cpu().numpy(), kd.float().cpu().numpy())
    assert np.array_equal(got.k_center().cpu().numpy(), ref["center_k"])
    assert np.array_equal(got.q_prep.cpu().numpy(), ref["q_prep"])
    assert np.array_equal(got.k_head_scale.cpu().numpy(), ref["k_head_scale"])
    assert np.array_equal(got.k8.cpu().numpy(), ref["k8"])
    assert np.array_equal(got.k_bias.cpu().numpy(), ref["k_bias"])
    assert ref["q_prep"][:, 1].min() < 0.6 and ref["q_prep"][:, 1].max() > 1.7  # the balance vector does something here
    assert np.abs(ref["k8"]).max() == 127
    # switches: no balancing / no centring
    plain = ops.i8_quantize_k(qd, kd, smooth=False, center=False)
    refp = O.i8_quantize_k(qd.float().cpu().numpy(), kd.float().cpu().numpy(), smooth=False, center=False)
    assert np.array_equal(plain.k8.cpu().numpy(), refp["k8"]) and float(plain.k_bias.abs().max()) == 0.0
    assert float(plain.q_prep[:, 1].min()) == float(plain.q_prep[:, 1].max()) == 1.0 and float(plain.q_prep[:, 0].abs().max()) == 0.0
    # strided input views (the projection buffer's (S, H*D) layout) give the same bytes
    kb = torch.empty((S, H, 128), dtype=dtype, device=dev())
    kb.copy_(kd.transpose(0, 1))
    qb = torch.empty((S, H, 128), dtype=dtype, device=dev())
    qb.copy_(qd.transpose(0, 1))
    strided = ops.i8_quantize_k(qb.transpose(0, 1), kb.transpose(0, 1))
    assert torch.equal(strided.k8, got.k8) and torch.equal(strided.k_bias, got.k_bias)


@pytest.mark.parametrize("T", [0, 40])
def test_i8_quantizer_segmented_layout_and_slot_groups_write_the_same_bytes(T):
    """the Ulysses receive layout: row r belongs to head slot (r // Sl) % Hl, text rows behind the video rows; one call,
    and one call per slot group, write what the plain (H, S + T, D) call writes for every head"""
    from vorta_amd import ops
    from vorta_amd.ulysses import UlyssesLayout
    dtype = torch.bfloat16
    Hl, P, Sl = 3, 4, 520
    S = P * Sl
    rng = np.random.default_rng(6 + T)
    q = rng.standard_normal((Hl, S + T, 128)) + rng.standard_normal((Hl, 1, 128))
    k = rng.standard_normal((Hl, S + T, 128)) * np.linspace(0.5, 2.0, 128) + rng.standard_normal((Hl, 1, 128))
    qd, kd = to_dev(q, dtype), to_dev(k, dtype)
    plain = ops.i8_quantize_k(qd, kd)
    lay = UlyssesLayout(Hl * P, S, T, 128, P, 1, dev(), dtype)
    assert lay.Hl == Hl and lay.Sl == Sl
    bufs = []
    for x in (qd, kd):
        b = lay.new_buffer()
        for src in range(P):
            b[src * Hl * Sl:(src + 1) * Hl * Sl] = x[:, src * Sl:(src + 1) * Sl].reshape(Hl * Sl, 128)
        for i in range(Hl):
            b[lay.rows_video + i * Sl: lay.rows_video + i * Sl + T] = x[i, S:]
        bufs.append(b)
    rm = lay.row_map.long()

    def fresh():
        return ops.I8Operands(torch.zeros((1, lay.rows_total, 128), dtype=torch.int8, device=dev()),
                              torch.zeros((1, lay.rows_total), dtype=torch.float32, device=dev()),
                              torch.zeros((Hl, 2, 128), dtype=torch.float32, device=dev()),
                              torch.zeros((Hl,), dtype=torch.float32, device=dev()),
                              torch.zeros(2 * Hl * 128 + Hl, dtype=torch.float32, device=dev()))

    def views(o):
        k8 = lay.head_view(o.k8[0])
        kb = o.k_bias[0].as_strided((Hl, lay.rows_total - (Hl - 1) * Sl), (Sl, 1))
        return k8[:, rm[:S + T]], kb[:, rm[:S + T]]

    kw = dict(heads=Hl, seg_len=Sl, tail_first=lay.rows_video, tail_len=T)
    one = ops.i8_quantize_k(bufs[0][None], bufs[1][None], out=fresh(), **kw)
    k8, kb = views(one)
    assert torch.equal(k8, plain.k8) and torch.equal(kb, plain.k_bias) and torch.equal(one.q_prep, plain.q_prep)
    assert torch.equal(one.k_head_scale, plain.k_head_scale)
    grouped = fresh()
    for g0, g1 in ((0, 2), (2, 3)):
        ops.i8_quantize_k(bufs[0][None], bufs[1][None], out=grouped, slots=(g0, g1), **kw)
    k8, kb = views(grouped)
    assert torch.equal(k8, plain.k8) and torch.equal(kb, plain.k_bias) and torch.equal(grouped.q_prep, plain.q_prep)
    assert torch.equal(grouped.k_head_scale, plain.k_head_scale)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("block_rows", [128, 256])
def test_i8_dense_ragged_vs_emulator(dtype, block_rows):
    from vorta_amd import ops
    rng = np.random.default_rng(1)
    H, Sq, Skv = 3, 333, 417
    q = rng.standard_normal((H, Skv, 128)) + 0.8 * rng.standard_normal((H, 1, 128))
    k, v = rng.standard_normal((H, Skv, 128)) + 0.5, rng.standard_normal((H, Skv, 128))
    n_kv, q_valid = 401, 300
    qd, kd = to_dev(q, dtype), to_dev(k, dtype)
    i8 = ops.i8_quantize_k(qd, kd)
    vdev = to_dev(v * np.linspace(0.05, 8.0, 128), dtype)
    v8, vd, _ = ops.fp8_quantize_v(vdev)
    out = torch.full((H, Sq, 128), 7.0, dtype=dtype, device=dev())
    ops.attn_fwd(qd[:, :Sq], i8.k8, v8, out, n_q=Sq, n_kv=n_kv, q_valid=q_valid, block_rows=block_rows, v_descale=vd, i8=i8)
    torch.cuda.synchronize()
    hooks = _hooks(qd, i8)
    ve, vde = _vdec(v8, vd)
    ref, exact, amb = np.zeros((H, Sq, 128)), np.zeros((H, Sq, 128)), np.zeros((H, Sq))
    for h in range(H):
        O.fp8_attn_launch(None, None, ve[h], ref[h], vde[h], n_q=Sq, n_kv=n_kv, q_valid=q_valid, ambiguous=amb[h],
                          wave_operands=hooks[h], p_mode="mx", defer=24.0)
        O.fp8_attn_launch(None, None, ve[h], exact[h], vde[h], n_q=Sq, n_kv=n_kv, q_valid=q_valid, round_p=False,
                          wave_operands=hooks[h])
    _check(out, ref, dtype, amb, _vmax(ve, vde))
    assert rel_fro(out.float().cpu().numpy(), exact) <= RELF_DIRECT
    assert torch.all(out[:, q_valid:] == 0)
    # and against exact attention on the 16-bit inputs: what int8 scores + e4m3 P, V cost together
    full = np.stack([O.dense_attention(qd[h, :Sq].double().cpu().numpy(), kd[h].double().cpu().numpy(),
                                       vdev[h].double().cpu().numpy(), kv_valid=n_kv, q_valid=q_valid) for h in range(H)])
    assert rel_fro(out.float().cpu().numpy(), full) <= 0.06


def test_i8_wave_on_the_head_centre_keeps_the_bias_term():
    """query rows that all equal the head's centre (qt = 0, abs-max 0) must give softmax(cq . (k - ck)) -- the bias
    term alone -- not a uniform row: against the emulator and against exact attention on the 16-bit inputs"""
    from vorta_amd import ops
    dtype = torch.bfloat16
    rng = np.random.default_rng(8)
    H, S = 2, 640
    k = rng.standard_normal((H, S, 128)) + 0.5
    v = rng.standard_normal((H, S, 128))
    q = rng.standard_normal((H, S, 128)) + 1.5 * rng.standard_normal((H, 1, 128))
    qd, kd = to_dev(q, dtype), to_dev(k, dtype)
    i8 = ops.i8_quantize_k(qd, kd)
    cq = i8.q_prep[:, 0].to(dtype)  # the centre, rounded to the input type: rows 64 ... 127 sit exactly on it
    qd[:, 64:128] = cq[:, None, :]
    i8.q_prep[:, 0] = cq.float()
    v8, vd, _ = ops.fp8_quantize_v(to_dev(v, dtype))
    out = torch.empty((H, S, 128), dtype=dtype, device=dev())
    ops.attn_fwd(qd, i8.k8, v8, out, n_q=S, n_kv=S, v_descale=vd, i8=i8)
    torch.cuda.synchronize()
    hooks = _hooks(qd, i8)
    ve, vde = _vdec(v8, vd)
    ref, amb = np.zeros((H, S, 128)), np.zeros((H, S))
    for h in range(H):
        O.fp8_attn_launch(None, None, ve[h], ref[h], vde[h], n_q=S, n_kv=S, ambiguous=amb[h], wave_operands=hooks[h],
                          p_mode="mx", defer=24.0)
    _check(out, ref, dtype, amb, _vmax(ve, vde))
    full = np.stack([O.dense_attention(qd[h].double().cpu().numpy(), kd[h].double().cpu().numpy(),
                                       to_dev(v, dtype)[h].double().cpu().numpy()) for h in range(H)])
    o = out.float().cpu().numpy()
    assert rel_fro(o[:, 64:128], full[:, 64:128]) < 0.08  # (a uniform softmax is off by far more)
    uniform = np.broadcast_to(to_dev(v, dtype).double().cpu().numpy().mean(1, keepdims=True), full.shape)
    assert rel_fro(uniform[:, 64:128], full[:, 64:128]) > 3 * rel_fro(o[:, 64:128], full[:, 64:128])


def test_i8_rescale_branch_long_keys_and_split_keys():
    """key norms grow along the sequence: the blocks climb tens of binades above the first one's maximum.  With the trigger at
    1 binade (`defer`) the reference point of every wave moves many times, with the default (24) never or once -- both against
    the emulator at the same trigger, and against each other (a threshold sweep: the branch changes scales, not values); then
    the same keys cut into 3 and 8 splits with the combine kernel"""
    from vorta_amd import ops
    dtype = torch.float16
    rng = np.random.default_rng(2)
    H, Sq, Skv = 2, 96, 2048
    q = rng.standard_normal((H, Skv, 128))
    k = rng.standard_normal((H, Skv, 128)) * np.linspace(0.3, 3.0, Skv)[None, :, None]
    v = rng.standard_normal((H, Skv, 128))
    qd, kd = to_dev(q, dtype), to_dev(k, dtype)
    i8 = ops.i8_quantize_k(qd, kd)
    v8, vd, _ = ops.fp8_quantize_v(to_dev(v, dtype))
    hooks = _hooks(qd, i8)
    ve, vde = _vdec(v8, vd)
    outs = {}
    for defer in (1.0, 24.0):
        for n_splits in (1, 3, 8):
            out = torch.empty((H, Sq, 128), dtype=dtype, device=dev())
            ops.attn_fwd(qd[:, :Sq], i8.k8, v8, out, n_q=Sq, n_kv=Skv, v_descale=vd, n_splits=n_splits, i8=i8,
                         fp8_opts={"defer": defer})
            ref, amb = np.zeros((H, Sq, 128)), np.zeros((H, Sq))
            moved = 0
            for h in range(H):
                O.fp8_attn_launch(None, None, ve[h], ref[h], vde[h], n_q=Sq, n_kv=Skv, n_splits=n_splits, ambiguous=amb[h],
                                  wave_operands=hooks[h], p_mode="mx", defer=defer)
            _check(out, ref, dtype, amb, _vmax(ve, vde))
            outs[(defer, n_splits)] = out.float().cpu().numpy()
        # the emulator's own reference points: with the low trigger they end far above the first block's maximum
    Qw, Kw = hooks[0](np.arange(32), np.arange(Skv))
    m_lo = O._fp8_flash_rows(Qw, Kw, ve[0], 0, Skv // 64, 5.0, 1.0, True, p_mode="mx")[2]
    m_hi = O._fp8_flash_rows(Qw, Kw, ve[0], 0, Skv // 64, 5.0, 24.0, True, p_mode="mx")[2]
    first = (Qw @ Kw[:64].T).max(1)
    assert (m_lo - first).min() > 4.0 and np.all(m_hi - first < m_lo - first + 1e-9)
    assert rel_fro(outs[(1.0, 1)], outs[(24.0, 1)]) < 2e-3  # scales moved, values did not


@pytest.mark.parametrize("block_rows", [128, 256])
def test_i8_tables_groups_duplicates_heads(block_rows):
    from vorta_amd import ops
    dtype = torch.bfloat16
    rng = np.random.default_rng(3)
    H, rows = 4, 700
    x = [rng.standard_normal((H, rows, 128)) for _ in range(3)]
    qd, kd = to_dev(x[0] + 0.7, dtype), to_dev(x[1], dtype)
    i8 = ops.i8_quantize_k(qd, kd)
    v8, vd, _ = ops.fp8_quantize_v(to_dev(x[2], dtype))
    n_q, glen, n_kv = 520, 200, 391  # 3 groups (200, 200, 120), own key list per group
    q_rows = rng.permutation(rows)[:n_q].astype(np.int32)
    kv_rows = np.stack([rng.permutation(rows)[:n_kv] for _ in range(3)]).astype(np.int32)
    free = np.setdiff1d(np.arange(rows), q_rows)
    dup = rng.permutation(free)[:2 * 60].reshape(60, 2).astype(np.int32)
    heads = torch.tensor([3, 0, 2], dtype=torch.int32, device=dev())
    count = torch.tensor([2], dtype=torch.int32, device=dev())
    out = torch.zeros((H, rows, 128), dtype=dtype, device=dev())
    ops.attn_fwd(qd, i8.k8, v8, out, head_list=heads, n_heads_dev=count, n_q=n_q, q_group_len=glen, n_kv=n_kv,
                 q_rows=torch.as_tensor(q_rows, device=dev()), kv_rows=torch.as_tensor(kv_rows, device=dev()),
                 kv_rows_stride_g=n_kv, dup_rows=torch.as_tensor(dup, device=dev()), n_dup_pos=60,
                 block_rows=block_rows, v_descale=vd, i8=i8)
    torch.cuda.synchronize()
    hooks = _hooks(qd, i8)
    ve, vde = _vdec(v8, vd)
    ref, amb = np.zeros((H, rows, 128)), np.zeros((H, rows))
    for h in (3, 0):
        O.fp8_attn_launch(None, None, ve[h], ref[h], vde[h], n_q=n_q, n_kv=n_kv, q_rows=q_rows, q_group_len=glen,
                          kv_rows=kv_rows, dup_rows=dup, n_dup_pos=60, ambiguous=amb[h], wave_operands=hooks[h],
                          p_mode="mx", defer=24.0)
    _check(out, ref, dtype, amb, _vmax(ve, vde))
    assert torch.all(out[2] == 0) and torch.all(out[1] == 0)


@pytest.mark.parametrize("model", ["hunyuan", "wan"])
@pytest.mark.parametrize("fused", [True, False])
def test_i8_routed_attention_vs_oracle(model, fused):
    """the whole routed op with precision "i8pv" -- fused grid and one launch per expert agree, device-resident routes give
    the same bytes, and every head sits close to the fp64 oracle on the 16-bit inputs"""
    from vorta_amd import ops
    from vorta_amd.routed import HeadRouting, RoutedGeometry, routed_attention
    dtype = torch.bfloat16
    latent, tile, window, group = (8, 12, 16), (2, 6, 8), (3, 3, 3), (2, 3, 2)
    S = latent[0] * latent[1] * latent[2]
    T, te = (256, 200) if model == "hunyuan" else (0, 0)
    H = 6
    rng = np.random.default_rng(11)
    q, k, v = (rng.standard_normal((1, H, S + T, 128)) for _ in range(3))
    experts = [0, 1, 2, 2, 1, 0]
    geom = RoutedGeometry(latent, tile, window, group, 0.5, dev())
    qd, kd, vd_ = to_dev(q, dtype), to_dev(k, dtype), to_dev(v, dtype)
    out = routed_attention(qd, kd, vd_, HeadRouting.from_expert_ids(experts, dev()), geom, model=model, text_len=T,
                           text_valid=te, fp8="i8pv", fused=fused)
    other = routed_attention(qd, kd, vd_, HeadRouting.from_expert_ids(experts, dev()), geom, model=model, text_len=T,
                             text_valid=te, fp8="i8pv", fused=not fused)
    assert float((out.float() - other.float()).abs().max()) <= 2e-2
    sc = torch.zeros((1, H, 3), device=dev())
    for h, e in enumerate(experts):
        sc[0, h, e] = 1.0
    _, lists, counts = ops.route_scores(sc, 0.3)
    out2 = routed_attention(qd, kd, vd_, HeadRouting.from_device(lists, counts), geom, model=model, text_len=T,
                            text_valid=te, fp8="i8pv", fused=fused)
    assert torch.equal(out2, out)
    from _util import rounded
    gi = O.group_info(latent, group, 0.5)
    full = O.routed_attention(rounded(q, dtype), rounded(k, dtype), rounded(v, dtype), np.array(experts), model=model,
                              latent=latent, tile=tile, window=window, gi=gi, t_text=T, t_eff=te)[0]
    o = out[0].float().cpu().numpy()
    rfs = [rel_fro(o[h], full[h]) for h in range(H)]
    print("i8pv routed vs the fp64 oracle on the 16-bit inputs, rel. Frobenius per head:", [round(x, 4) for x in rfs])
    assert max(rfs) < 0.06, rfs
    if T:
        assert torch.all(out[0, :, S + te:] == 0)


# the targets of VERDICT r03 item 2 (dB over max|x| of the bf16 result), not the measured values
I8_GATES = {"white": 40.0, "common3": 40.0, "student_t3": 40.0, "smooth": 40.0, "peaked": 40.0, "outlier_w": 40.0,
            "outlier_w_common": 39.0}
# the relative Frobenius error against the bf16 kernels, which a PSNR over max|x| hides on heavy tails.  Up
# to ABI 6 the probabilities' RANGE cost 0.09-0.13 on the common-part and smooth families and 0.14-0.21 on Student-t; with one
# scale per (query row, 32 keys) six families sit at 0.012-0.064 (target 0.08).  Student-t(3) keeps 0.10-0.19: what is left there
# is the int8 SCORE under one key scale per head and one query scale per 32 rows (the abs-max of 10^7 t(3) samples is ~250
# sigma: the bulk rounds to 0 or +-1; CPU study with exact scores + the same probabilities + e4m3 v: 0.027) -- a per-row key
# scale costs an instruction per score (round 4), so that family is gated at what the format gives
# (profiles/r05_mx_probabilities.txt); inputs like it belong on precision "fp8pv" (16-bit scores)
I8_REL_GATE = 0.08
I8_REL_GATES = {"student_t3": 0.20}


@pytest.mark.parametrize("precision", ["i8pv", "auto8"])
@pytest.mark.parametrize("geometry", ["wan14b-81f", "hunyuan-129f"])
def test_i8_operator_psnr_on_every_input_family(geometry, precision):
    """gate (ii): every expert, every input family, precision "i8pv" against the bf16 kernels on the same bf16 inputs: PSNR
    over max|x| >= 40 dB, relative Frobenius error <= 0.08 (Student-t: 0.20, see I8_REL_GATES); PSNR over the 99.9th percentile
    of |x| is printed beside them.  precision "auto8" (heavy-tailed heads take 16-bit scores, ABI 8): <= 0.08 on EVERY family."""
    from _fp8_inputs import NAMES, families, psnr, robust_psnr
    from vorta_amd.routed import HeadRouting, RoutedGeometry, routed_attention
    dtype = torch.bfloat16
    if geometry == "wan14b-81f":
        latent, tile, window, group, model, T, te = (21, 45, 80), (7, 9, 8), (3, 3, 3), (3, 3, 2), "wan", 0, 0
    else:
        latent, tile, window, group, model, T, te = (33, 45, 80), (11, 9, 8), (3, 3, 3), (3, 3, 2), "hunyuan", 256, 96
    S = latent[0] * latent[1] * latent[2]
    geom = RoutedGeometry(latent, tile, window, group, 0.5, dev())
    routing = HeadRouting.from_expert_ids([0, 1, 2], dev())
    gen = torch.Generator(device=dev()).manual_seed(1234)
    kw = dict(model=model, text_len=T, text_valid=te)
    experts = ["full", "coreset", "sliding"]
    failures = []
    for key, q, k, v in families(latent, 3, T, gen, dev()):
        q16, k16, v16 = (x.to(dtype)[None].contiguous() for x in (q, k, v))
        ref = routed_attention(q16, k16, v16, routing, geom, **kw)
        out = routed_attention(q16, k16, v16, routing, geom, fp8=precision, **kw)
        torch.cuda.synchronize()
        assert torch.isfinite(out.float()).all(), key
        table = {experts[h]: psnr(out[0, h, :S + te], ref[0, h, :S + te]) + (robust_psnr(out[0, h, :S + te], ref[0, h, :S + te]),)
                 for h in range(3)}
        print(f"{precision} vs bf16 [{geometry}] {NAMES[key]}: " + ", ".join(f"{n} {a:.1f} / {b:.1f} / p99.9 {d:.1f} dB rel {c:.3f}"
                                                                      for n, (a, b, c, d) in table.items()))
        for n, (p_range, p_peak, rel, p_rob) in table.items():
            if p_peak < I8_GATES[key] or rel > (I8_REL_GATE if precision == "auto8" else I8_REL_GATES.get(key, I8_REL_GATE)):
                failures.append((geometry, key, n, round(p_peak, 1), round(rel, 3)))
    assert not failures, failures


# ---------------------------------------------------------------------------------------------------------------- "auto8"
def _auto8_case(dev, fams, H_each=1, seed=5):
    """q, k, v whose heads come from different input families (H_each heads per family), one small Wan geometry"""
    from _fp8_inputs import families
    latent = (9, 18, 16)  # 2 592 tokens: a Student-t(3) head's abs-max is ~60 sigma there (the int8 keys' rms ~2 counts)
    gen = torch.Generator(device=dev).manual_seed(seed)
    parts = {name: (q, k, v) for name, q, k, v in families(latent, H_each, 0, gen, dev) if name in fams}
    q, k, v = (torch.cat([parts[f][i] for f in fams], 0).to(torch.bfloat16).unsqueeze(0).contiguous() for i in range(3))
    return latent, q, k, v


def test_auto8_flags_heavy_tailed_heads_and_each_head_equals_the_kernel_it_was_given_to():
    """vorta_i8_tail_flags + vorta_split_heads (ABI 8): of the heads [white, Student-t(3), smooth, Student-t(3), outlier weights]
    the two heavy-tailed ones are flagged; under "auto8" every head's output is BIT FOR BIT what the precision it was routed to
    writes for it ("i8pv" for the unflagged, "fp8pv" for the flagged), whatever expert it belongs to."""
    from vorta_amd import ops
    from vorta_amd.routed import HeadRouting, RoutedGeometry, routed_attention
    dev = torch.device("cuda")
    fams = ["white", "student_t3", "smooth", "student_t3", "outlier_w"]
    latent, q, k, v = _auto8_case(dev, fams)
    H = len(fams)
    i8 = ops.i8_quantize_k(q[0], k[0])
    flags = ops.i8_tail_flags(i8.k8)
    assert flags.tolist() == [0, 1, 0, 1, 0], flags.tolist()
    # the split keeps the order and honours a device-side count
    hl = torch.tensor([4, 3, 1, 0, 2], dtype=torch.int32, device=dev)
    n_dev = torch.tensor([4], dtype=torch.int32, device=dev)
    a, b = ops.split_heads(flags, hl, 5, n_dev)
    assert a["head_list"][:int(a["n_heads_dev"])].tolist() == [4, 0] and b["head_list"][:int(b["n_heads_dev"])].tolist() == [3, 1]
    a, b = ops.split_heads(flags, None, 5)
    assert a["head_list"][:int(a["n_heads_dev"])].tolist() == [0, 2, 4] and b["head_list"][:int(b["n_heads_dev"])].tolist() == [1, 3]
    geom = RoutedGeometry(latent, tile=(3, 6, 8), window=(3, 3, 3), group=(3, 3, 2), rate=0.5, device=dev)
    for experts in ([0, 0, 0, 0, 0], [0, 1, 2, 1, 2], [2, 2, 1, 0, 1]):
        routing = HeadRouting.from_expert_ids(experts, dev)
        outs = {p: routed_attention(q, k, v, routing, geom, model="wan", fp8=p) for p in ("auto8", "i8pv", "fp8pv")}
        for h, f in enumerate(flags.tolist()):
            want = outs["fp8pv" if f else "i8pv"][0, h]
            assert torch.equal(outs["auto8"][0, h], want), (experts, h, f)
    # the form the sequence-parallel path uses: converted views + 16-bit keys + flags handed in; flags through a row map
    v8, vd, _ = ops.fp8_quantize_v(v[0])
    o_views = routed_attention(q, k, v, routing, geom, model="wan", fp8=False, fp8_views=(q[0], i8.k8, v8, vd, i8, k[0], flags))
    assert torch.equal(o_views, outs["auto8"])
    perm = torch.randperm(q.shape[2], device=dev)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(perm.numel(), device=dev)
    shuffled = i8.k8[:, inv].contiguous()  # row perm[t] of the shuffled keys holds token t
    assert ops.i8_tail_flags(shuffled, row_map=perm.to(torch.int32)).tolist() == flags.tolist()
    # device-resident routes (counts only the device knows): the same bits
    routing = HeadRouting.from_expert_ids([0, 1, 2, 1, 2], dev)
    rd = HeadRouting.from_device(routing.lists, torch.tensor(routing.counts_host, dtype=torch.int32, device=dev))
    o_dev = routed_attention(q, k, v, rd, geom, model="wan", fp8="auto8")
    assert torch.equal(o_dev, routed_attention(q, k, v, routing, geom, model="wan", fp8="auto8"))


def test_auto8_relative_error_on_every_input_family():
    """the gate "i8pv" misses on one family: relative Frobenius error <= 0.08 against the 16-bit kernels on ALL seven input
    families under "auto8" (Student-t(3) heads take 16-bit scores), >= 40 dB; which heads were flagged is printed"""
    from _fp8_inputs import families, psnr
    from vorta_amd import ops
    from vorta_amd.routed import HeadRouting, RoutedGeometry, routed_attention
    dev = torch.device("cuda")
    latent, H = (9, 18, 16), 3
    gen = torch.Generator(device=dev).manual_seed(11)
    geom = RoutedGeometry(latent, tile=(3, 6, 8), window=(3, 3, 3), group=(3, 3, 2), rate=0.5, device=dev)
    routing = HeadRouting.from_expert_ids([0, 1, 2], dev)
    for name, q, k, v in families(latent, H, 0, gen, dev):
        q, k, v = (x.to(torch.bfloat16).unsqueeze(0).contiguous() for x in (q, k, v))
        ref = routed_attention(q, k, v, routing, geom, model="wan", fp8=False).float()
        got = routed_attention(q, k, v, routing, geom, model="wan", fp8="auto8").float()
        flags = ops.i8_tail_flags(ops.i8_quantize_k(q[0], k[0]).k8).tolist()
        rel = float((got - ref).norm() / ref.norm())
        db = psnr(got, ref)[1]  # over max |ref|, as the other gates
        print(f"auto8 vs bf16, {name}: flagged heads {flags}, rel {rel:.4f}, PSNR {db:.1f} dB")
        assert rel <= I8_REL_GATE and db >= 40.0, (name, rel, db)
        assert flags == ([1, 1, 1] if name == "student_t3" else [0, 0, 0]), (name, flags)


@pytest.mark.parametrize("seed", range(10))
def test_routed_attention_random_configs_8bit(seed):
    """The routed op under "fp8pv", "i8pv" and "auto8" on random small geometries -- odd tile / window / group shapes, experts
    without heads, text lengths, host- or device-resident routes, fused or serial launches: finite, padded text rows zero,
    every head within 0.15 relative of the 16-bit kernels (an indexing slip is off by O(1)), and every head of "auto8" bit for
    bit one of the other two."""
    from vorta_amd.routed import HeadRouting, RoutedGeometry, routed_attention
    rng = np.random.default_rng(9100 + seed)
    group = [(2, 3, 2), (3, 1, 2), (1, 2, 2), (2, 2, 1)][int(rng.integers(0, 4))]
    tile = tuple(int(rng.integers(1, 4)) for _ in range(3))
    latent = tuple(int(np.lcm(g, t)) * int(rng.integers(1, 4)) for g, t in zip(group, tile))
    window = tuple(int(rng.choice([1, 3, 5])) for _ in range(3))
    model = ("hunyuan", "wan")[int(rng.integers(0, 2))]
    T = int(rng.integers(1, 20)) if model == "hunyuan" else 0
    te = int(rng.integers(1, T + 1)) if T else 0
    H = int(rng.integers(1, 7))
    experts = rng.integers(0, 3, size=H)
    Sv = latent[0] * latent[1] * latent[2]
    q, k, v = (to_dev(rng.standard_normal((1, H, Sv + T, 128)), torch.bfloat16) for _ in range(3))
    geom = RoutedGeometry(latent, tile, window, group, 0.5, dev())
    host = HeadRouting.from_expert_ids(experts.tolist(), dev())
    routing = host if rng.integers(0, 2) else HeadRouting.from_device(
        host.lists, torch.tensor(host.counts_host, dtype=torch.int32, device=dev()))
    kw = dict(model=model, text_len=T, text_valid=te, fused=bool(rng.integers(0, 2)))
    desc = dict(seed=seed, model=model, latent=latent, tile=tile, window=window, group=group, T=T, te=te, experts=experts.tolist())
    ref = routed_attention(q, k, v, routing, geom, fp8=False, **kw).float()
    outs = {p: routed_attention(q, k, v, routing, geom, fp8=p, **kw) for p in ("fp8pv", "i8pv", "auto8")}
    torch.cuda.synchronize()
    for p, out in outs.items():
        o = out.float()
        assert torch.isfinite(o).all(), (p, desc)
        if T:
            assert (o[0, :, Sv + te:] == 0).all(), (p, desc)
        for h in range(H):
            rel = float((o[0, h] - ref[0, h]).norm() / ref[0, h].norm().clamp_min(1e-6))
            assert rel <= 0.15, (p, h, rel, desc)
    for h in range(H):
        assert torch.equal(outs["auto8"][0, h], outs["i8pv"][0, h]) or torch.equal(outs["auto8"][0, h], outs["fp8pv"][0, h]), (h, desc)
