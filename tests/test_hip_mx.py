"""GPU parity of the MIXED-precision attention (csrc/attn_fwd_mx.hip; vorta_attn_fwd_fp8 with ext->flags bit1): scores
in 16 bits, P V in e4m3.  Same gates as the all-e4m3 path (tests/test_hip_fp8.py):
  (i)   kernel vs the oracle's emulator on the SAME operands -- q pre-multiplied by scale * log2(e) and re-rounded to the
        16-bit type exactly as the kernel does, k as it is, v decoded from the e4m3 bytes -- with the probabilities
        rounded to e4m3 at the kernel's reference points: the 16-bit tolerances plus the emulator's midpoint slack;
  (i')  kernel vs exact attention on the same operands (no probability rounding): rel. Frobenius <= 3e-2;
  (ii)  operator PSNR against the 16-bit kernels on every input family of tests/_fp8_inputs.py, both geometries, every
        expert: >= 40 dB over max|x| -- the gate the all-e4m3 path cannot hold on peaked logits."""
import math

import numpy as np
import pytest
import torch

from oracle import vorta_oracle as O

pytestmark = pytest.mark.gpu

from _util import dev, rel_fro, to_dev  # noqa: E402
from test_hip_fp8 import RELF_PACK, _check, _vmax  # noqa: E402

MX = dict(p_mode="rne_mx", defer=24.0)  # the emulator's mode for this kernel since ABI 7 (one scale per query row and 32 keys)


def _operands(qd, kd, v8, vd, dtype, scale=None):
    """what the kernel multiplies: (q', k, v8 decoded, v_descale) as float64 arrays"""
    sc = np.float32(1.0 / math.sqrt(qd.shape[-1]) if scale is None else scale) * np.float32(1.4426950408889634)
    qe = (qd.float() * float(sc)).to(dtype).double().cpu().numpy()
    return qe, kd.double().cpu().numpy(), O.e4m3_decode(v8.cpu().numpy()), vd.cpu().numpy().astype(np.float64)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("block_rows", [128, 256])
def test_mx_dense_ragged_vs_emulator(dtype, block_rows):
    from vorta_amd import ops
    rng = np.random.default_rng(1)
    H, Sq, Skv = 3, 333, 417
    q, k, v = rng.standard_normal((H, Sq, 128)), rng.standard_normal((H, Skv, 128)), rng.standard_normal((H, Skv, 128))
    n_kv, q_valid = 401, 300
    qd, kd = to_dev(q, dtype), to_dev(k, dtype)
    v8, vd, _ = ops.fp8_quantize_v(to_dev(v * np.linspace(0.05, 8.0, 128), dtype))
    out = torch.full((H, Sq, 128), 7.0, dtype=dtype, device=dev())
    ops.attn_fwd(qd, kd, v8, out, n_q=Sq, n_kv=n_kv, q_valid=q_valid, block_rows=block_rows, v_descale=vd)
    torch.cuda.synchronize()
    qe, ke, ve, vde = _operands(qd, kd, v8, vd, dtype)
    ref, exact, amb = np.zeros((H, Sq, 128)), np.zeros((H, Sq, 128)), np.zeros((H, Sq))
    for h in range(H):
        O.fp8_attn_launch(qe[h], ke[h], ve[h], ref[h], vde[h], n_q=Sq, n_kv=n_kv, q_valid=q_valid, ambiguous=amb[h], **MX)
        O.fp8_attn_launch(qe[h], ke[h], ve[h], exact[h], vde[h], n_q=Sq, n_kv=n_kv, q_valid=q_valid, round_p=False)
    _check(out, ref, dtype, amb, _vmax(ve, vde))
    assert rel_fro(out.float().cpu().numpy(), exact) <= RELF_PACK
    assert torch.all(out[:, q_valid:] == 0)
    # v's e4m3 copy and scales are what the quantiser of the all-e4m3 path writes for the same v
    f8 = ops.fp8_quantize_qkv(qd[:, :1].expand(-1, Skv, -1).contiguous(), kd, to_dev(v * np.linspace(0.05, 8.0, 128), dtype))
    assert torch.equal(f8.v, v8) and torch.equal(f8.v_descale, vd)


def test_mx_rescale_branch_long_keys_and_split_keys():
    """key norms grow along the sequence: the blocks climb tens of binades above the first one's maximum.  With the trigger at
    1 binade (`defer`) the reference point of every wave moves many times, with the default (24) never or once -- both against
    the emulator at the same trigger and against each other (the branch changes scales, not values); then the same keys cut
    into 3 and 8 splits with the combine kernel"""
    from vorta_amd import ops
    dtype = torch.float16
    rng = np.random.default_rng(2)
    H, Sq, Skv = 2, 96, 2048
    q = rng.standard_normal((H, Skv, 128))
    k = rng.standard_normal((H, Skv, 128)) * np.linspace(0.3, 3.0, Skv)[None, :, None]
    v = rng.standard_normal((H, Skv, 128))
    qd, kd = to_dev(q, dtype), to_dev(k, dtype)
    v8, vd, _ = ops.fp8_quantize_v(to_dev(v, dtype))
    qe, ke, ve, vde = _operands(qd, kd, v8, vd, dtype)
    outs = {}
    for defer in (1.0, 24.0):
        for n_splits in (1, 3, 8):
            out = torch.empty((H, Sq, 128), dtype=dtype, device=dev())
            ops.attn_fwd(qd[:, :Sq], kd, v8, out, n_q=Sq, n_kv=Skv, v_descale=vd, n_splits=n_splits, fp8_opts={"defer": defer})
            ref, amb = np.zeros((H, Sq, 128)), np.zeros((H, Skv))
            for h in range(H):
                O.fp8_attn_launch(qe[h], ke[h], ve[h], ref[h], vde[h], n_q=Sq, n_kv=Skv, n_splits=n_splits, ambiguous=amb[h],
                                  p_mode="rne_mx", defer=defer)
            _check(out, ref, dtype, amb[:, :Sq], _vmax(ve, vde))
            outs[(defer, n_splits)] = out.float().cpu().numpy()
    assert rel_fro(outs[(1.0, 1)], outs[(24.0, 1)]) < 2e-3  # scales moved, values did not


@pytest.mark.parametrize("block_rows", [128, 256])
def test_mx_tables_groups_duplicates_heads(block_rows):
    from vorta_amd import ops
    dtype = torch.bfloat16
    rng = np.random.default_rng(3)
    H, rows = 4, 700
    x = [rng.standard_normal((H, rows, 128)) for _ in range(3)]
    qd, kd = to_dev(x[0], dtype), to_dev(x[1], dtype)
    v8, vd, _ = ops.fp8_quantize_v(to_dev(x[2], dtype))
    n_q, glen, n_kv = 520, 200, 391  # 3 groups (200, 200, 120), own key list per group
    q_rows = rng.permutation(rows)[:n_q].astype(np.int32)
    kv_rows = np.stack([rng.permutation(rows)[:n_kv] for _ in range(3)]).astype(np.int32)
    free = np.setdiff1d(np.arange(rows), q_rows)
    dup = rng.permutation(free)[:2 * 60].reshape(60, 2).astype(np.int32)
    heads = torch.tensor([3, 0, 2], dtype=torch.int32, device=dev())
    count = torch.tensor([2], dtype=torch.int32, device=dev())
    out = torch.zeros((H, rows, 128), dtype=dtype, device=dev())
    ops.attn_fwd(qd, kd, v8, out, head_list=heads, n_heads_dev=count, n_q=n_q, q_group_len=glen, n_kv=n_kv,
                 q_rows=torch.as_tensor(q_rows, device=dev()), kv_rows=torch.as_tensor(kv_rows, device=dev()),
                 kv_rows_stride_g=n_kv, dup_rows=torch.as_tensor(dup, device=dev()), n_dup_pos=60,
                 block_rows=block_rows, v_descale=vd)
    torch.cuda.synchronize()
    qe, ke, ve, vde = _operands(qd, kd, v8, vd, dtype)
    ref, amb = np.zeros((H, rows, 128)), np.zeros((H, rows))
    for h in (3, 0):
        O.fp8_attn_launch(qe[h], ke[h], ve[h], ref[h], vde[h], n_q=n_q, n_kv=n_kv, q_rows=q_rows, q_group_len=glen,
                          kv_rows=kv_rows, dup_rows=dup, n_dup_pos=60, ambiguous=amb[h], **MX)
    _check(out, ref, dtype, amb, _vmax(ve, vde))
    assert torch.all(out[2] == 0) and torch.all(out[1] == 0)


@pytest.mark.parametrize("model", ["hunyuan", "wan"])
@pytest.mark.parametrize("fused", [True, False])
def test_mx_routed_attention_vs_oracle(model, fused):
    """the whole routed op with precision "fp8pv" -- fused grid and one launch per expert agree, device-resident routes
    give the same bytes, and every head sits close to the fp64 oracle on the 16-bit inputs (the whole cost of the path:
    e4m3 P and V)"""
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
                           text_valid=te, fp8="fp8pv", fused=fused)
    other = routed_attention(qd, kd, vd_, HeadRouting.from_expert_ids(experts, dev()), geom, model=model, text_len=T,
                             text_valid=te, fp8="fp8pv", fused=not fused)
    assert float((out.float() - other.float()).abs().max()) <= 2e-2
    sc = torch.zeros((1, H, 3), device=dev())
    for h, e in enumerate(experts):
        sc[0, h, e] = 1.0
    _, lists, counts = ops.route_scores(sc, 0.3)
    out2 = routed_attention(qd, kd, vd_, HeadRouting.from_device(lists, counts), geom, model=model, text_len=T,
                            text_valid=te, fp8="fp8pv", fused=fused)
    assert torch.equal(out2, out)
    from _util import rounded
    gi = O.group_info(latent, group, 0.5)
    full = O.routed_attention(rounded(q, dtype), rounded(k, dtype), rounded(v, dtype), np.array(experts), model=model,
                              latent=latent, tile=tile, window=window, gi=gi, t_text=T, t_eff=te)[0]
    o = out[0].float().cpu().numpy()
    rfs = [rel_fro(o[h], full[h]) for h in range(H)]
    print("fp8pv routed vs the fp64 oracle on the 16-bit inputs, rel. Frobenius per head:", [round(x, 4) for x in rfs])
    assert max(rfs) < 0.05, rfs
    if T:
        assert torch.all(out[0, :, S + te:] == 0)


# relative Frobenius error against the bf16 kernels: 16-bit scores + probabilities with one scale per query
# row and 32 keys hold it on EVERY family, Student-t(3) included (0.13-0.15 there up to ABI 6: the probabilities' range)
MX_REL_GATE = 0.08


@pytest.mark.parametrize("geometry", ["wan14b-81f", "hunyuan-129f"])
def test_mx_operator_psnr_on_every_input_family(geometry):
    """gate (ii): every expert, every input family, precision "fp8pv" against the bf16 kernels on the same bf16 inputs:
    >= 40 dB over max|x| of the 16-bit result -- including the peaked-softmax and outlier-channel families on which the
    all-e4m3 path sits at 36 / 29 / 21 dB"""
    from _fp8_inputs import NAMES, families, psnr, robust_psnr
    from vorta_amd.routed import HeadRouting, RoutedGeometry, routed_attention
    dtype = torch.bfloat16
    failures = []
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
    for key, q, k, v in families(latent, 3, T, gen, dev()):
        q16, k16, v16 = (x.to(dtype)[None].contiguous() for x in (q, k, v))
        ref = routed_attention(q16, k16, v16, routing, geom, **kw)
        out = routed_attention(q16, k16, v16, routing, geom, fp8="fp8pv", **kw)
        torch.cuda.synchronize()
        assert torch.isfinite(out.float()).all(), key
        table = {experts[h]: psnr(out[0, h, :S + te], ref[0, h, :S + te]) + (robust_psnr(out[0, h, :S + te], ref[0, h, :S + te]),)
                 for h in range(3)}
        print(f"fp8pv vs bf16 [{geometry}] {NAMES[key]}: " + ", ".join(f"{n} {a:.1f} / {b:.1f} / p99.9 {d:.1f} dB rel {c:.3f}"
                                                                       for n, (a, b, c, d) in table.items()))
        for n, (p_range, p_peak, rel, p_rob) in table.items():
            if p_peak < 40.0 or rel > MX_REL_GATE:
                failures.append((geometry, key, n, round(p_peak, 1), round(rel, 3)))
    assert not failures, failures


@pytest.mark.parametrize("block_rows", [128, 256])
def test_8bit_kernels_with_one_two_three_key_blocks(block_rows):
    """The wave-role loops (e4m3, mixed, int8-score kernels) lag P V two blocks behind the scores, start with a step that has no
    P V and end with a drain: every short key count -- 1 ... 3 blocks of 64, whole and ragged -- against exact attention on the
    16-bit operands (a pipeline slip gives garbage, not a rounding error), padded query rows zero, untouched rows untouched."""
    from vorta_amd import ops
    rng = np.random.default_rng(7)
    H, Sq, dtype = 2, 300, torch.bfloat16
    for n_kv in (1, 40, 64, 65, 127, 128, 129, 192, 193, 200):
        q, k, v = rng.standard_normal((H, Sq, 128)), rng.standard_normal((H, n_kv, 128)), rng.standard_normal((H, n_kv, 128))
        qd, kd, vd16 = to_dev(q, dtype), to_dev(k, dtype), to_dev(v, dtype)
        want = O.dense_attention(qd.double().cpu().numpy(), kd.double().cpu().numpy(), vd16.double().cpu().numpy(),
                                 kv_valid=n_kv, q_valid=Sq - 20)
        v8, vd, _ = ops.fp8_quantize_v(vd16)
        i8 = ops.i8_quantize_k(qd[:, :n_kv].contiguous(), kd)  # (q is only sampled for the centre and balance statistics)
        cases = {"fp8pv": dict(q=qd, k=kd, v=v8, v_descale=vd), "i8pv": dict(q=qd, k=i8.k8, v=v8, v_descale=vd, i8=i8)}
        for name, c in cases.items():
            out = torch.full((H, Sq + 4, 128), 7.0, dtype=dtype, device=dev())
            ops.attn_fwd(c.pop("q"), c.pop("k"), c.pop("v"), out, n_q=Sq, n_kv=n_kv, q_valid=Sq - 20, block_rows=block_rows, **c)
            torch.cuda.synchronize()
            got = out.float().cpu().numpy()
            assert np.isfinite(got).all() and (got[:, Sq:] == 7.0).all() and (got[:, Sq - 20:Sq] == 0).all(), (name, n_kv)
            assert rel_fro(got[:, :Sq - 20], want[:, :Sq - 20]) <= 0.08, (name, n_kv, rel_fro(got[:, :Sq - 20], want[:, :Sq - 20]))
