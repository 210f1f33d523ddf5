"""GPU: the geometries of BASELINE.json's configs at full size, checked through size-independent properties and
sampled rows against the oracle (the oracle cannot hold an S x S score matrix at these sizes).

  configs[0]  Wan-2.1 1.3B 49x320x512  (13,20,32)  S =   8 320  native (dense) attention, 12 heads
  configs[1]  Wan-2.1 1.3B 81x480x832  (21,30,52)  S =  32 760  routed, tile (7,6,4), coreset (3,3,2)
  configs[2]  HunyuanVideo 129x720x1280 (33,45,80) S = 118 800  routed: tests/test_hip_experts.py::test_full_size_*
  configs[4]  Wan-2.1 14B 81x720x1280  (21,45,80)  S =  75 600  routed, tile (7,9,8), coreset (3,3,2), 40 heads (bf16, and
              one rank of eight through the e4m3 kernels: test_config4_one_rank_of_eight_fp8_at_full_size)
"""
import os

import numpy as np
import pytest
import torch

from oracle import vorta_oracle as O
from _util import ATOL_SAME, check, dev

pytestmark = pytest.mark.gpu
WINDOW = (3, 3, 3)


def _rand(shape, seed, dtype):
    gen = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=gen).to(dtype).to(dev())


def test_config0_wan13b_49f_native_attention():
    from vorta_amd.patch import wan_pixel2token
    from vorta_amd.routed import dense_attention
    latent = wan_pixel2token((49, 320, 512))
    assert latent == (13, 20, 32)
    S, H, dtype = 13 * 20 * 32, 12, torch.bfloat16
    q, k, v = (_rand((1, H, S, 128), s, dtype) for s in (1, 2, 3))
    out = dense_attention(q, k, v)
    for h in (0, 7):  # two full heads against the oracle
        ref = O.dense_attention(q[0, h].double().cpu().numpy(), k[0, h].double().cpu().numpy(), v[0, h].double().cpu().numpy())
        check(out[0, h], ref, dtype)
        # the north star's image-level bar, applied to the operator: PSNR >= 40 dB against the float64 result
        got = out[0, h].double().cpu().numpy()
        psnr = 20.0 * np.log10(np.abs(ref).max() / np.sqrt(np.mean((got - ref) ** 2)))
        assert psnr >= 40.0, psnr
    # every head: softmax rows sum to one (constant V reproduced)
    const = _rand((1, H, 1, 128), 4, dtype)
    out2 = dense_attention(q, k, const.expand(1, H, S, 128).contiguous())
    assert (out2.float() - const.float()).abs().max().item() <= 2e-2


def _verify_samples(model, geom, q, k, v, experts, rows_of, dtype, seed, text=(0, 0), layout=None):
    """Per expert: sampled query rows against the oracle run on the key / query lists the kernels used (the tables are
    bit-exact vs the reference at small size), plus the structural properties.  q,k,v: (1,H,S+T,128) in TOKEN order;
    rows_of(h, token_ids) -> the kernel's output rows of head h for those tokens (whatever layout it wrote them in)."""
    from vorta_amd import ops
    S, (T, te) = geom.S, text
    latent, group = geom.latent, geom.group
    gen = torch.Generator().manual_seed(seed)
    tol = ATOL_SAME[dtype]
    f64 = lambda t: t.double().cpu().numpy()
    got = lambda h, ids: rows_of(h, torch.as_tensor(ids).to(dev()).long()).float().cpu().numpy()
    h_full, h_low, h_sl = (experts.index(e) for e in (0, 1, 2))
    # full expert: sampled rows vs dense oracle
    rows = torch.randint(0, S, (32,), generator=gen)
    ref = O.dense_attention(f64(q[0, h_full, rows]), f64(k[0, h_full, :S + te]), f64(v[0, h_full, :S + te]))
    assert np.abs(got(h_full, rows) - ref).max() <= tol
    # coreset expert: rows of the packed sequence vs dense oracle over the kept keys; dropped margins == their centre
    keep_q, drop_q = ops.coreset_select(q[0, h_low:h_low + 1], latent, group, geom.n_keep, tail_first=S, n_tail=T)
    keep_k = keep_q if model == "wan" else ops.coreset_select(k[0, h_low:h_low + 1], latent, group, geom.n_keep,
                                                              tail_first=S, n_tail=te, want_drop=False)[0]
    kk = keep_k[0, :geom.S_low + te].long()
    pos = torch.randint(0, geom.S_low, (32,), generator=gen)
    qr = keep_q[0].cpu()[pos].long()
    ref = O.dense_attention(f64(q[0, h_low, qr]), f64(k[0, h_low, kk]), f64(v[0, h_low, kk]))
    assert np.abs(got(h_low, qr) - ref).max() <= tol
    centres = keep_q[0, :geom.G].long()
    sub = torch.randint(0, geom.G, (4096,), generator=gen).to(dev())
    dr = drop_q[0][sub].long()
    assert torch.equal(rows_of(h_low, dr.reshape(-1)).reshape(dr.shape + (128,)),
                       rows_of(h_low, centres[sub])[:, None].expand(-1, dr.shape[-1], -1))
    # sliding expert: sampled rows vs dense oracle over the tile's key list (token-order tables of the plain geometry)
    plain = geom if geom.row_map is None else None
    if plain is None:
        from vorta_amd.routed import RoutedGeometry
        plain = RoutedGeometry(geom.latent, geom.tile, geom.window, geom.group, geom.rate, dev())
    q_rows, kv_rows, _ = plain.sta_tables(te)
    pos = torch.randint(0, S, (24,), generator=gen)
    for p in pos.tolist():
        keys = kv_rows[p // geom.tok].long()
        r = int(q_rows[p])
        ref = O.dense_attention(f64(q[0, h_sl, r:r + 1]), f64(k[0, h_sl, keys]), f64(v[0, h_sl, keys]))
        assert np.abs(got(h_sl, [r]) - ref).max() <= tol
    if T:  # text queries: valid ones attend every valid key (all three experts), padded ones are exactly zero
        trow = torch.arange(S, S + te)
        for h, keys in ((h_full, torch.arange(S + te)), (h_sl, torch.arange(S + te)),
                        (h_low, kk.cpu())):
            ref = O.dense_attention(f64(q[0, h, trow]), f64(k[0, h, keys]), f64(v[0, h, keys]))
            assert np.abs(got(h, trow) - ref).max() <= tol, h
            assert torch.all(rows_of(h, torch.arange(S + te, S + T, device=dev())) == 0)
    _lists_vs_oracle(model, geom, q, k, h_low, seed, text, layout)


def _lists_vs_oracle(model, geom, q, k, h_low, seed, text=(0, 0), layout=None):
    """The lists the kernels READ THROUGH at full size against the oracle's own index code -- nothing here comes from the
    library except the lists under test (`_verify_samples` feeds the oracle the kernel's lists).
      * coreset (coreset_select.py:98-113): 512 random window groups -> O.coreset_similarity on those groups -> centre, kept
        and dropped margins index for index wherever the similarity gaps exceed fp32 noise (Q matching; Hunyuan: K too);
      * sliding tile (sliding_attn_flex.py:93-128, tile.py:7-41): q_rows == O.tile_major_order; 32 random query tiles: the key
        rows of the tile == the tokens of the tiles O.sta_window_tiles lets it see (+ the valid text); the same for 32
        workgroups of the merged launch table (query tiles of equal key lists share a group).
    `layout` = (q_view, k_view, row_map): the lists of a sequence-parallel rank are rows of its receive buffers."""
    from vorta_amd import ops
    S, (T, te) = geom.S, text
    rng = np.random.default_rng(seed)
    rm = None if layout is None else layout[2].long().cpu().numpy()
    to_rows = (lambda tok: tok) if rm is None else (lambda tok: rm[tok])
    gi = O.group_info(geom.latent, geom.group, geom.rate)
    G, nk = gi.n_groups, gi.n_keep_margin
    assert G == geom.G and nk == geom.n_keep
    idx = np.sort(rng.choice(G, size=min(512, G), replace=False))
    sub = O.GroupInfo(center=gi.center[idx], margin=gi.margin[idx], n_keep_margin=nk)
    for which, x, n_tail in (("q", q, T), ("k", k, te)) if model == "hunyuan" else (("q", q, T),):
        if layout is None:
            keep, drop = ops.coreset_select(x[0, h_low:h_low + 1], geom.latent, geom.group, nk, tail_first=S, n_tail=n_tail)
        else:
            xv = layout[0] if which == "q" else layout[1]
            keep, drop = ops.coreset_select(xv[h_low:h_low + 1], geom.latent, geom.group, nk, tail_first=S, n_tail=n_tail,
                                            row_map=layout[2])
        keep, drop = keep[0].cpu().numpy(), drop[0].cpu().numpy()
        xh = x[0, h_low, :S].double().cpu().numpy()[None, None]
        sims = O.coreset_similarity(xh, sub)[0, 0]  # (n, g-1), float64 on the same rounded inputs
        order = np.argsort(sims, axis=-1, kind="stable")
        kept_ref = np.take_along_axis(sub.margin, order[:, :nk], axis=1)
        drop_ref = np.take_along_axis(sub.margin, order[:, nk:], axis=1)
        clear = np.diff(np.sort(sims, axis=-1), axis=-1).min(-1) > 1e-5
        assert clear.mean() > 0.9, (which, clear.mean())
        assert np.array_equal(keep[:G][idx], to_rows(sub.center[:, 0])), which
        assert np.array_equal(keep[G:G + G * nk].reshape(G, nk)[idx][clear], to_rows(kept_ref)[clear]), which
        assert np.array_equal(drop[idx][clear], to_rows(drop_ref)[clear]), which
        assert np.array_equal(keep[G + G * nk:], to_rows(np.arange(S, S + n_tail))), which
    # ---- sliding tile ----
    q_rows, kv_rows, n_kv = geom.sta_tables(te)
    q_rows, kv_rows = q_rows.cpu().numpy(), kv_rows.cpu().numpy()
    perm = O.tile_major_order(geom.latent, geom.tile)
    assert np.array_equal(q_rows, to_rows(perm))
    sees = O.sta_window_tiles(geom.latent, geom.tile, geom.window)
    tok, n_tiles = geom.tok, S // geom.tok
    text_rows = to_rows(np.arange(S, S + te))

    def keys_of(ti):
        return np.concatenate([to_rows(perm[j * tok:(j + 1) * tok]) for j in np.nonzero(sees[ti])[0]] + [text_rows])

    for ti in rng.choice(n_tiles, size=min(32, n_tiles), replace=False):
        want = keys_of(ti)
        assert n_kv == want.size and np.array_equal(np.sort(kv_rows[ti]), np.sort(want)), ti
    # the launch the routed op submits: merged groups of query tiles with equal key lists
    q_m, lists, n_kv_m, table, n_lists = geom.sta_launch_tables(te, 256)
    q_m, lists, table = q_m.cpu().numpy(), lists.cpu().numpy(), table.cpu().numpy()
    assert n_kv_m == n_kv and np.array_equal(np.sort(q_m), np.sort(q_rows)) and lists.shape[0] == n_lists
    covered = np.zeros(S, dtype=np.int32)
    for g, p0, p1 in table:
        covered[p0:p1] += 1
    assert (covered == 1).all()  # every query position in exactly one workgroup
    inv = np.empty(S + T if rm is None else int(rm.max()) + 1, dtype=np.int64)
    inv[q_rows] = np.arange(S) // tok  # row -> its tile (tile-major position / tokens per tile)
    for b in rng.choice(table.shape[0], size=min(32, table.shape[0]), replace=False):
        g, p0, p1 = table[b]
        tiles = np.unique(inv[q_m[p0:p1]])
        for ti in tiles:
            assert np.array_equal(np.sort(lists[g]), np.sort(keys_of(ti))), (b, ti)


def _sampled_expert_checks(model, latent, tile, group, H, experts, dtype, seed, text=(0, 0), fp8=False):
    """Routed op at full size, then `_verify_samples`."""
    from vorta_amd.routed import HeadRouting, RoutedGeometry, routed_attention
    S = latent[0] * latent[1] * latent[2]
    T, te = text
    q, k, v = (_rand((1, H, S + T, 128), seed + i, dtype) for i in range(3))
    geom = RoutedGeometry(latent, tile, WINDOW, group, 0.5, dev())
    out = routed_attention(q, k, v, HeadRouting.from_expert_ids(experts, dev()), geom, model=model, text_len=T,
                           text_valid=te)
    _verify_samples(model, geom, q, k, v, list(experts), lambda h, ids: out[0, h, ids], dtype, seed, text)
    return out


def test_config1_wan13b_81f_routed():
    from vorta_amd.patch import wan_pixel2token
    assert wan_pixel2token((81, 480, 832)) == (21, 30, 52)
    experts = [0, 1, 2, 2, 1, 0, 1, 2, 0, 0, 2, 1]
    _sampled_expert_checks("wan", (21, 30, 52), (7, 6, 4), (3, 3, 2), 12, experts, torch.bfloat16, seed=100)


def test_config4_wan14b_81f_geometry_routed_bf16():
    from vorta_amd.patch import wan_pixel2token
    assert wan_pixel2token((81, 720, 1280)) == (21, 45, 80)
    experts = [0, 1, 2, 1]  # 4 of the 40 heads are enough to exercise the geometry at S = 75 600
    _sampled_expert_checks("wan", (21, 45, 80), (7, 9, 8), (3, 3, 2), 4, experts, torch.bfloat16, seed=200)


def test_reference_native_geometry_hunyuan_117f():
    """The authors' own benchmark shape (vorta/constants.py:8-12; tile (6,9,8), coreset (2,3,2))."""
    experts = [0, 1, 2]
    _sampled_expert_checks("hunyuan", (30, 45, 80), (6, 9, 8), (2, 3, 2), 3, experts, torch.float16, seed=300,
                           text=(256, 77))


def test_config2_hunyuan_129f_routed_fp16_as_benched():
    """BASELINE configs[2] exactly as bench.py runs its first layer: H = 24, S = 118 800 + 256/96 text, fp16, the
    'uniform' mix with bench.py's layer-0 head -> expert draw, the experts as ONE fused grid."""
    import bench
    from vorta_amd import ops
    from vorta_amd.routed import HeadRouting, RoutedGeometry, routed_attention
    cfg = bench.CONFIGS["hunyuan-129f"]
    experts = [int(e) for e in bench.layer_experts(cfg, "uniform", 0)]
    assert len(experts) == 24 and [experts.count(e) for e in range(3)] == [8, 8, 8]
    dtype = torch.float16
    S, T, te = 33 * 45 * 80, cfg["text"], cfg["text_valid"]
    q, k, v = (_rand((1, 24, S + T, 128), 400 + i, dtype) for i in range(3))
    geom = RoutedGeometry(cfg["latent"], cfg["tile"], cfg["window"], cfg["group"], cfg["rate"], dev())
    tl = ops.Timeline()
    ops.set_timeline(tl)
    try:
        out = routed_attention(q, k, v, HeadRouting.from_expert_ids(experts, dev()), geom, model="hunyuan", text_len=T,
                               text_valid=te)
    finally:
        ops.set_timeline(None)
    assert [r[1] for r in tl.records] == ["attn_fwd_multi_kernel<_Float16>"]  # the kernel the bench line reports
    _verify_samples("hunyuan", geom, q, k, v, experts, lambda h, ids: out[0, h, ids], dtype, 401, (T, te))
    # a second head of every expert (the last one routed to it)
    last = [len(experts) - 1 - experts[::-1].index(e) for e in range(3)]
    swapped = list(experts)
    for e in range(3):
        first = experts.index(e)
        swapped[first], swapped[last[e]] = -1, e  # make `index(e)` find the last head of the expert
    _verify_samples("hunyuan", geom, q, k, v, swapped, lambda h, ids: out[0, h, ids], dtype, 402, (T, te))


def test_config3_one_rank_of_eight_at_full_size():
    """BASELINE configs[3]: HunyuanVideo 129f under 8-way Ulysses.  What ONE rank holds after the exchange -- 3 heads x the
    whole sequence, laid out as 8 received chunks of (3, S/8, D) + the replicated text -- is built here without any
    communication (as test_zero_copy_ulysses_layout_on_one_gpu does at S = 384) and attended in place through `row_map`;
    sampled rows of every expert are checked against the oracle."""
    from vorta_amd.routed import HeadRouting, RoutedGeometry, routed_attention
    from vorta_amd.ulysses import UlyssesLayout
    dtype = torch.float16
    H, P, rank = 24, 8, 5
    latent, tile, group = (33, 45, 80), (11, 9, 8), (3, 3, 2)
    S, T, te = 33 * 45 * 80, 256, 96
    lay = UlyssesLayout(H, S, T, 128, P, rank, dev(), dtype)
    Hl, Sl = lay.Hl, lay.Sl
    assert (Hl, Sl) == (3, 14850)
    q, k, v = (_rand((1, Hl, S + T, 128), 500 + i, dtype) for i in range(3))  # the rank's heads, token order
    bufs = []
    for x in (q, k, v):
        b = lay.new_buffer().zero_()
        for src in range(P):  # chunk received from rank `src`: its S/8 tokens of each of my 3 heads
            b[src * Hl * Sl:(src + 1) * Hl * Sl] = x[0, :, src * Sl:(src + 1) * Sl].reshape(Hl * Sl, 128)
        for i in range(Hl):
            b[lay.rows_video + i * Sl: lay.rows_video + i * Sl + T] = x[0, i, S:]
        bufs.append(b)
    obuf = lay.new_buffer().zero_()
    geom = RoutedGeometry(latent, tile, WINDOW, group, 0.5, dev(), row_map=lay.row_map)
    experts = [2, 0, 1]
    routed_attention(*(lay.head_view(b) for b in bufs), HeadRouting.from_expert_ids(experts, dev()), geom,
                     model="hunyuan", text_len=T, text_valid=te, out=lay.head_view(obuf))
    ov, rm = lay.head_view(obuf), lay.row_map.long()
    views = [lay.head_view(b) for b in bufs]
    _verify_samples("hunyuan", geom, q, k, v, experts, lambda h, ids: ov[h][rm[ids]], dtype, 501, (T, te),
                    layout=(views[0], views[1], lay.row_map))


def test_config4_one_rank_of_eight_fp8_at_full_size():
    """BASELINE configs[4]: Wan-2.1 14B 81x720x1280, Ulysses over 8 GPUs, fp8 contractions.  One rank's share -- 5 of the
    40 heads over the whole S = 75 600 sequence in the zero-copy receive layout, converted to e4m3 ONCE per buffer
    (UlyssesLayout.fp8_views) -- against the bf16 kernels on the same layout: gate (ii) of tests/test_hip_fp8.py."""
    import math
    from vorta_amd.routed import HeadRouting, RoutedGeometry, routed_attention
    from vorta_amd.ulysses import UlyssesLayout
    dtype = torch.bfloat16
    H, P, rank = 40, 8, 3
    latent, tile, group = (21, 45, 80), (7, 9, 8), (3, 3, 2)
    S = 21 * 45 * 80
    lay = UlyssesLayout(H, S, 0, 128, P, rank, dev(), dtype)
    Hl, Sl = lay.Hl, lay.Sl
    assert (Hl, Sl) == (5, 9450)
    bufs = []
    for i in range(3):
        b = lay.new_buffer()
        b[:lay.rows_video] = _rand((lay.rows_video, 128), 600 + i, dtype)
        bufs.append(b)
    geom = RoutedGeometry(latent, tile, WINDOW, group, 0.5, dev(), row_map=lay.row_map)
    route = HeadRouting.from_expert_ids([0, 1, 2, 1, 2], dev())
    views = [lay.head_view(b) for b in bufs]
    ref, out = lay.new_buffer(), lay.new_buffer()
    routed_attention(*views, route, geom, model="wan", out=lay.head_view(ref), fp8=False)
    q8, k8, v8, vd, f8 = lay.fp8_views(bufs)
    assert f8.q.shape == (1, lay.rows_total, 128) and vd.shape == (Hl, 128)
    routed_attention(*views, route, geom, model="wan", out=lay.head_view(out), fp8=False, fp8_views=(q8, k8, v8, vd))
    torch.cuda.synchronize()
    rm = lay.row_map.long()
    a, b = lay.head_view(out)[:, rm[:S]].float(), lay.head_view(ref)[:, rm[:S]].float()  # (Hl, S, D) in token order
    assert not torch.isnan(a).any()
    for i in range(Hl):
        mse = ((a[i] - b[i]) ** 2).mean().item()
        psnr = 10 * math.log10((b[i].max() - b[i].min()).item() ** 2 / mse)
        assert psnr >= 40.0, (i, psnr)


def test_hunyuan_129f_fp8_with_text_and_biased_keys_at_full_size():
    """The e4m3 path on the headline geometry (S = 118 800, text 256 / 96 valid) through the Ulysses receive layout of a
    rank of 8 (3 heads, per-head scales and key centres from the segmented quantiser) -- one head per expert, keys with a
    3-sigma common component: operator PSNR >= 40 dB per expert against the fp16 kernels, valid text rows included; the
    padded text rows are exactly zero."""
    import math
    from vorta_amd.routed import HeadRouting, RoutedGeometry, routed_attention
    from vorta_amd.ulysses import UlyssesLayout
    dtype = torch.float16
    H, P, rank = 24, 8, 2
    latent, tile, group = (33, 45, 80), (11, 9, 8), (3, 3, 2)
    S, T, te = 33 * 45 * 80, 256, 96
    lay = UlyssesLayout(H, S, T, 128, P, rank, dev(), dtype)
    Hl, Sl = lay.Hl, lay.Sl
    bufs = []
    for i in range(3):
        b = lay.new_buffer()
        x = _rand((lay.rows_total, 128), 700 + i, dtype)
        if i == 1:  # keys: every head slot gets its own common component
            seg = (torch.arange(lay.rows_total, device=dev()) // Sl) % Hl
            bias = 3.0 * _rand((Hl, 128), 710, torch.float32)
            x = (x.float() + bias[seg]).to(dtype)
        b[:lay.rows_video] = x[:lay.rows_video]
        for s in range(Hl):
            b[lay.rows_video + s * Sl: lay.rows_video + s * Sl + T] = x[lay.rows_video + s * Sl: lay.rows_video + s * Sl + T]
        bufs.append(b)
    geom = RoutedGeometry(latent, tile, WINDOW, group, 0.5, dev(), row_map=lay.row_map)
    route = HeadRouting.from_expert_ids([0, 1, 2], dev())
    views = [lay.head_view(b) for b in bufs]
    ref, out = lay.new_buffer(), lay.new_buffer()
    routed_attention(*views, route, geom, model="hunyuan", text_len=T, text_valid=te, out=lay.head_view(ref), fp8=False)
    q8, k8, v8, vd, f8 = lay.fp8_views(bufs)
    c = f8.k_center()
    assert vd.shape == (Hl, 128) and float((c - bias).abs().max()) < 0.5  # each slot's own centre, near its own bias
    routed_attention(*views, route, geom, model="hunyuan", text_len=T, text_valid=te, out=lay.head_view(out), fp8=False,
                     fp8_views=(q8, k8, v8, vd))
    torch.cuda.synchronize()
    rm = lay.row_map.long()
    a, b = lay.head_view(out)[:, rm].float(), lay.head_view(ref)[:, rm].float()  # (Hl, S + T, D) in token order
    assert not torch.isnan(a).any()
    assert (a[:, S + te:] == 0).all() and (b[:, S + te:] == 0).all()
    table = {}
    for i, name in enumerate(("full", "coreset", "sliding-tile")):
        for part, sl in (("video", slice(0, S)), ("text", slice(S, S + te))):
            mse = ((a[i, sl] - b[i, sl]) ** 2).mean().item()
            table[f"{name}/{part}"] = 10 * math.log10((b[i, sl].max() - b[i, sl].min()).item() ** 2 / mse)
    print("fp8 vs fp16, Hunyuan-129f geometry, keys with a 3-sigma common component (PSNR over data range, dB):",
          {n: round(p, 2) for n, p in table.items()})
    assert min(table.values()) >= 40.0, table


def test_hunyuan_129f_fp8pv_one_rank_of_eight_at_full_size():
    """Precision "fp8pv" (16-bit scores, e4m3 P V) on the headline geometry through the Ulysses receive layout of a rank of
    8: v is converted the way the exchange does it -- per-(head, channel) abs-max taken shard by shard (P calls that only
    raise the maximum) and one conversion with the whole-sequence scales into the e4m3 receive buffer -- q and k are read as
    they landed.  Keys with a 3-sigma common component and a peaked softmax (q, k x 1.5): >= 55 dB over max|x| per expert on
    the video rows, >= 42 dB on the valid text rows, against the fp16 kernels; padded text rows exactly zero."""
    import math
    from vorta_amd import ops
    from vorta_amd.routed import HeadRouting, RoutedGeometry, routed_attention
    from vorta_amd.ulysses import UlyssesLayout
    dtype = torch.float16
    H, P, rank = 24, 8, 2
    latent, tile, group = (33, 45, 80), (11, 9, 8), (3, 3, 2)
    S, T, te = 33 * 45 * 80, 256, 96
    lay = UlyssesLayout(H, S, T, 128, P, rank, dev(), dtype)
    Hl, Sl = lay.Hl, lay.Sl
    bufs = []
    for i in range(3):
        b = lay.new_buffer()
        x = _rand((lay.rows_total, 128), 800 + i, dtype)
        if i < 2:
            x = (x.float() * 1.5).to(dtype)
        if i == 1:
            seg = (torch.arange(lay.rows_total, device=dev()) // Sl) % Hl
            x = (x.float() + (3.0 * _rand((Hl, 128), 810, torch.float32))[seg]).to(dtype)
        b[:lay.rows_video] = x[:lay.rows_video]
        for s_ in range(Hl):
            b[lay.rows_video + s_ * Sl: lay.rows_video + s_ * Sl + T] = x[lay.rows_video + s_ * Sl: lay.rows_video + s_ * Sl + T]
        bufs.append(b)
    geom = RoutedGeometry(latent, tile, WINDOW, group, 0.5, dev(), row_map=lay.row_map)
    route = HeadRouting.from_expert_ids([0, 1, 2], dev())
    views = [lay.head_view(b) for b in bufs]
    ref, out = lay.new_buffer(), lay.new_buffer()
    routed_attention(*views, route, geom, model="hunyuan", text_len=T, text_valid=te, out=lay.head_view(ref), fp8=False)
    # v: abs-max chunk by chunk (what the P senders contribute through the MAX all-reduce), then one conversion per chunk
    vb = bufs[2]
    amax = torch.zeros((Hl, 128), dtype=torch.float32, device=dev())
    chunks = [vb[j * Hl * Sl:(j + 1) * Hl * Sl].view(Hl, Sl, 128) for j in range(P)]
    texts = vb[lay.rows_video:].as_strided((Hl, T, 128), (Sl * 128, 128, 1))
    for c in chunks + [texts]:
        ops.fp8_v_absmax(c, amax)
    v8 = torch.zeros((lay.rows_total, 128), dtype=torch.uint8, device=dev())
    vd = torch.empty((Hl, 128), dtype=torch.float32, device=dev())
    for j, c in enumerate(chunks):
        ops.fp8_v_convert(c, amax, v8[j * Hl * Sl:(j + 1) * Hl * Sl].view(Hl, Sl, 128), v_descale=vd)
    ops.fp8_v_convert(texts, amax, v8[lay.rows_video:].as_strided((Hl, T, 128), (Sl * 128, 128, 1)))
    routed_attention(*views, route, geom, model="hunyuan", text_len=T, text_valid=te, out=lay.head_view(out), fp8=False,
                     fp8_views=(views[0], views[1], lay.head_view(v8), vd))
    torch.cuda.synchronize()
    rm = lay.row_map.long()
    a, b = lay.head_view(out)[:, rm].float(), lay.head_view(ref)[:, rm].float()
    assert not torch.isnan(a).any()
    assert (a[:, S + te:] == 0).all() and (b[:, S + te:] == 0).all()
    table = {}
    for i, name in enumerate(("full", "coreset", "sliding-tile")):
        for part, sl in (("video", slice(0, S)), ("text", slice(S, S + te))):
            mse = ((a[i, sl] - b[i, sl]) ** 2).mean().item()
            table[f"{name}/{part}"] = 10 * math.log10(b[i, sl].abs().max().item() ** 2 / mse)
    print("fp8pv vs fp16, Hunyuan-129f geometry, rank of 8, peaked softmax + key bias (PSNR over max|x|, dB):",
          {n: round(p, 2) for n, p in table.items()})
    # (the 96 valid text queries attend every key: a flat softmax over 118 896 keys, the e4m3 noise of P and V averages least)
    assert all(v >= (55.0 if n.endswith("video") else 42.0) for n, v in table.items()), table


@pytest.mark.parametrize("precision", ["fp8", "fp8pv"])
@pytest.mark.parametrize("config", ["wan14b-81f", "hunyuan-129f"])
def test_fp8_headline_sizes_sampled_waves_vs_oracle_emulator(config, precision):
    """The e4m3 kernels at the sizes that matter against the ORACLE (not against the 16-bit kernels): one head per expert
    over the whole sequence of BASELINE configs[4] (Wan-2.1 14B 81x720x1280, S = 75 600) and of the headline geometry
    (HunyuanVideo 129f, S = 118 800 + 256 text rows, 96 valid), the routed op as the processors call it (fused grid),
    then sampled waves -- 32 consecutive query positions, the unit that shares reference-point decisions -- of every
    launch restated by oracle.fp8_attn_launch on the kernel's own e4m3 operands, rounding point for rounding point
    (same tolerance rule as tests/test_hip_fp8.py at small sizes).  precision "fp8pv": the mixed kernel (16-bit scores,
    e4m3 P V) the same way, on the operands tests/test_hip_mx.py gives the emulator."""
    import test_hip_fp8 as F
    from vorta_amd import ops
    from vorta_amd.routed import HeadRouting, RoutedGeometry, routed_attention
    if config == "wan14b-81f":
        model, latent, tile, group, T, te, dtype = "wan", (21, 45, 80), (7, 9, 8), (3, 3, 2), 0, 0, torch.bfloat16
    else:
        model, latent, tile, group, T, te, dtype = "hunyuan", (33, 45, 80), (11, 9, 8), (3, 3, 2), 256, 96, torch.float16
    S = latent[0] * latent[1] * latent[2]
    experts = [0, 1, 2]
    q, k, v = (_rand((1, 3, S + T, 128), 900 + i, dtype) for i in range(3))
    geom = RoutedGeometry(latent, tile, WINDOW, group, 0.5, dev())
    if precision == "fp8":
        f8 = ops.fp8_quantize_qkv(q[0], k[0], v[0], center_k=True)
        out = routed_attention(q, k, v, HeadRouting.from_expert_ids(experts, dev()), geom, model=model, text_len=T,
                               text_valid=te, fp8=True, fp8_operands=f8)
        torch.cuda.synchronize()
        q8, k8, v8, vd = F._decoded(f8)
    else:
        import test_hip_mx as M
        out = routed_attention(q, k, v, HeadRouting.from_expert_ids(experts, dev()), geom, model=model, text_len=T,
                               text_valid=te, fp8="fp8pv")
        torch.cuda.synchronize()
        v8_, vd_, _ = ops.fp8_quantize_v(v[0])
        q8, k8, v8, vd = M._operands(q[0], k[0], v8_, vd_, dtype)
    em = dict(M.MX) if precision == "fp8pv" else {}  # the emulator's mode: the mixed kernel scales its probabilities per tile
    N = S + T
    ref, amb = np.zeros((3, N, 128)), np.full((3, N), np.nan)
    gen = np.random.default_rng(7)

    def some(n_groups_lens, n):
        """n random (group, first position) pairs of a launch whose groups have the given lengths"""
        picks = set()
        while len(picks) < n:
            g = int(gen.integers(len(n_groups_lens)))
            picks.add((g, 32 * int(gen.integers(-(-n_groups_lens[g] // 32)))))
        return lambda g, w0: (g, w0) in picks

    # full expert: one group of S + T positions (the last waves hold the text rows)
    nq = S + T
    picks = some([nq], 5)
    last = ((S + te - 1) // 32) * 32  # the wave with the last valid text row / last video rows
    O.fp8_attn_launch(q8[0], k8[0], v8[0], ref[0], vd[0], n_q=nq, n_kv=S + te, q_valid=S + te, ambiguous=amb[0],
                      wave_filter=lambda g, w0: picks(g, w0) or w0 == last, **em)
    # coreset expert
    hl = torch.tensor([1], dtype=torch.int32, device=dev())
    keep_q, drop_q = ops.coreset_select(q[0], geom.latent, geom.group, geom.n_keep, tail_first=S, n_tail=T, head_list=hl)
    keep_k = keep_q if model == "wan" else ops.coreset_select(k[0], geom.latent, geom.group, geom.n_keep, tail_first=S,
                                                              n_tail=te, head_list=hl, want_drop=False)[0]
    nql = geom.S_low + T
    picks = some([nql], 5)
    O.fp8_attn_launch(q8[1], k8[1], v8[1], ref[1], vd[1], n_q=nql, n_kv=geom.S_low + te, q_valid=geom.S_low + te,
                      q_rows=keep_q[0].cpu().numpy(), kv_rows=keep_k[0].cpu().numpy(), dup_rows=drop_q[0].cpu().numpy(),
                      n_dup_pos=geom.G, ambiguous=amb[1], wave_filter=lambda g, w0: picks(g, w0) or w0 == 0, **em)
    # sliding-tile expert (query tiles of equal key lists merged, as vorta_amd/routed.py launches them)
    q_rows, kv_rows, n_kv, table, n_lists = geom.sta_launch_tables(te, 256)
    qr, kr, tb = q_rows.cpu().numpy(), kv_rows.cpu().numpy(), table.cpu().numpy()
    bounds = [(int(tb[tb[:, 0] == g, 1].min()), int(tb[tb[:, 0] == g, 2].max())) for g in range(n_lists)]
    O.fp8_attn_launch(q8[2], k8[2], v8[2], ref[2], vd[2], n_q=S, n_kv=n_kv, q_rows=qr, kv_rows=kr, q_group_bounds=bounds,
                      ambiguous=amb[2], wave_filter=some([b[1] - b[0] for b in bounds], 12), **em)
    if T:
        # inside the fused grid the text launch of the sliding expert is unsplit (vorta_amd/routed.py FUSED_TEXT_SPLITS)
        O.fp8_attn_launch(q8[2], k8[2], v8[2], ref[2], vd[2], n_q=T, q_row_offset=S, q_valid=te, n_kv=S + te,
                          n_splits=1, ambiguous=amb[2], wave_filter=lambda g, w0: w0 in (0, 64), **em)
    o = out[0].float().cpu().numpy()
    vmax = F._vmax(v8, vd)
    for h in range(3):
        sel = ~np.isnan(amb[h])
        assert sel.sum() >= 32 * 5, (h, sel.sum())
        F._check(torch.from_numpy(o[h][sel]), ref[h][sel], dtype, amb[h][sel], vmax)
    if T:
        assert torch.all(out[0, :, S + te:] == 0)


@pytest.mark.parametrize("config", ["wan14b-81f", "hunyuan-129f"])
def test_i8pv_headline_sizes_sampled_waves_vs_oracle_emulator(config):
    """precision "i8pv" (int8 scores, e4m3 P V) at the sizes that matter against the ORACLE: one head per
    expert over the whole sequence of BASELINE configs[4] (S = 75 600) and of the headline geometry (S = 118 800 + 256 text
    rows, 96 valid), the routed op as the processors call it (fused grid: the kernel with wave roles), then sampled waves of
    every launch restated by oracle.fp8_attn_launch on the kernel's own operands -- the wave's query conversion, the int8
    keys and their seeds (O.i8_wave_operands), the probabilities' bytes written as the kernel writes them."""
    import test_hip_fp8 as F
    import test_hip_i8 as I
    from vorta_amd import ops
    from vorta_amd.routed import HeadRouting, RoutedGeometry, routed_attention
    if config == "wan14b-81f":
        model, latent, tile, group, T, te, dtype = "wan", (21, 45, 80), (7, 9, 8), (3, 3, 2), 0, 0, torch.bfloat16
    else:
        model, latent, tile, group, T, te, dtype = "hunyuan", (33, 45, 80), (11, 9, 8), (3, 3, 2), 256, 96, torch.float16
    S = latent[0] * latent[1] * latent[2]
    q, k, v = (_rand((1, 3, S + T, 128), 910 + i, dtype) for i in range(3))
    geom = RoutedGeometry(latent, tile, WINDOW, group, 0.5, dev())
    v8_, vd_, _ = ops.fp8_quantize_v(v[0])
    i8 = ops.i8_quantize_k(q[0], k[0])
    out = routed_attention(q, k, v, HeadRouting.from_expert_ids([0, 1, 2], dev()), geom, model=model, text_len=T,
                           text_valid=te, fp8="i8pv", fp8_operands=((v8_, vd_, _), i8))
    torch.cuda.synchronize()
    hooks = I._hooks(q[0], i8)
    ve, vde = I._vdec(v8_, vd_)
    N = S + T
    ref, amb = np.zeros((3, N, 128)), np.full((3, N), np.nan)
    gen = np.random.default_rng(9)

    def some(lens, n):
        picks = set()
        while len(picks) < n:
            g = int(gen.integers(len(lens)))
            picks.add((g, 32 * int(gen.integers(-(-lens[g] // 32)))))
        return lambda g, w0: (g, w0) in picks

    kw = dict(p_mode="mx", defer=24.0)
    more = int(os.environ.get("VORTA_TEST_MORE_WAVES", "1"))  # an extended run samples this many times more waves
    nq = S + T
    picks = some([nq], 4 * more)
    last = ((S + te - 1) // 32) * 32
    O.fp8_attn_launch(None, None, ve[0], ref[0], vde[0], n_q=nq, n_kv=S + te, q_valid=S + te, ambiguous=amb[0],
                      wave_filter=lambda g, w0: picks(g, w0) or w0 == last, wave_operands=hooks[0], **kw)
    hl = torch.tensor([1], dtype=torch.int32, device=dev())
    keep_q, drop_q = ops.coreset_select(q[0], geom.latent, geom.group, geom.n_keep, tail_first=S, n_tail=T, head_list=hl)
    keep_k = keep_q if model == "wan" else ops.coreset_select(k[0], geom.latent, geom.group, geom.n_keep, tail_first=S,
                                                              n_tail=te, head_list=hl, want_drop=False)[0]
    nql = geom.S_low + T
    picks = some([nql], 4 * more)
    O.fp8_attn_launch(None, None, ve[1], ref[1], vde[1], n_q=nql, n_kv=geom.S_low + te, q_valid=geom.S_low + te,
                      q_rows=keep_q[0].cpu().numpy(), kv_rows=keep_k[0].cpu().numpy(), dup_rows=drop_q[0].cpu().numpy(),
                      n_dup_pos=geom.G, ambiguous=amb[1], wave_filter=lambda g, w0: picks(g, w0) or w0 == 0,
                      wave_operands=hooks[1], **kw)
    q_rows, kv_rows, n_kv, table, n_lists = geom.sta_launch_tables(te, 256)
    qr, kr, tb = q_rows.cpu().numpy(), kv_rows.cpu().numpy(), table.cpu().numpy()
    bounds = [(int(tb[tb[:, 0] == g, 1].min()), int(tb[tb[:, 0] == g, 2].max())) for g in range(n_lists)]
    O.fp8_attn_launch(None, None, ve[2], ref[2], vde[2], n_q=S, n_kv=n_kv, q_rows=qr, kv_rows=kr, q_group_bounds=bounds,
                      ambiguous=amb[2], wave_filter=some([b[1] - b[0] for b in bounds], 10 * more), wave_operands=hooks[2], **kw)
    if T:
        O.fp8_attn_launch(None, None, ve[2], ref[2], vde[2], n_q=T, q_row_offset=S, q_valid=te, n_kv=S + te, n_splits=1,
                          ambiguous=amb[2], wave_filter=lambda g, w0: w0 in (0, 64), wave_operands=hooks[2], **kw)
    o = out[0].float().cpu().numpy()
    vmax = F._vmax(ve, vde)
    for h in range(3):
        sel = ~np.isnan(amb[h])
        assert sel.sum() >= 32 * 4, (h, sel.sum())
        F._check(torch.from_numpy(o[h][sel]), ref[h][sel], dtype, amb[h][sel], vmax)
    if T:
        assert torch.all(out[0, :, S + te:] == 0)
