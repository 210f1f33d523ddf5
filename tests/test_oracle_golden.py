"""Pin the CPU oracle (oracle/vorta_oracle.py) to golden vectors produced by running the reference
(tools/gen_goldens.py).  CPU only."""
import numpy as np
import pytest

from oracle import vorta_oracle as O

LATENT, TILE, WINDOW, GROUP = (8, 6, 8), (2, 3, 4), (3, 3, 3), (2, 3, 2)
S = 8 * 6 * 8


def _t(a):
    return tuple(int(x) for x in a)


# ---------------------------------------------------------------- G1
@pytest.mark.parametrize("tag", ["4x6x4", "8x6x8", "9x6x8_g18", "9x7x9_crop", "8x6x8_r075"])
def test_group_info(golden, tag):
    g = golden("g1_group_info")
    gi = O.group_info(_t(g[f"{tag}_latent"]), _t(g[f"{tag}_window"]), float(g[f"{tag}_rate"]))
    assert np.array_equal(gi.center, g[f"{tag}_center"])
    assert np.array_equal(gi.margin, g[f"{tag}_margin"])
    assert gi.n_keep_margin == int(g[f"{tag}_num_unpooled"])


# ---------------------------------------------------------------- G2
def test_pool_unpool(golden):
    g = golden("g2_pool_unpool")
    gi = O.group_info(LATENT, GROUP, 0.5)
    x = g["x"].astype(np.float64)
    kept, dropped = O.coreset_match(x, gi)
    assert np.array_equal(kept, g["unpooled_argsort"])
    assert np.array_equal(dropped, g["pooled_argsort"])
    pooled = O.coreset_pool(x, gi, kept)
    np.testing.assert_allclose(pooled, g["pooled"], rtol=0, atol=0)
    np.testing.assert_array_equal(O.coreset_pool(g["y"].astype(np.float64), gi, kept), g["pooled_y"])
    np.testing.assert_array_equal(O.coreset_unpool(pooled, gi, kept, dropped), g["unpooled"])
    # the index form used by the HIP path describes the same gather / scatter
    keep_rows, drop_rows = O.coreset_row_lists(gi, kept, dropped)
    np.testing.assert_array_equal(np.take_along_axis(x, keep_rows[..., None], axis=2), g["pooled"])
    assert keep_rows.shape[-1] + drop_rows.shape[-1] * drop_rows.shape[-2] == S


# ---------------------------------------------------------------- G3
@pytest.mark.parametrize("tag", ["hy_text", "wan_notext", "narrow_t", "win531"])
def test_sta_mask(golden, tag):
    g = golden("g3_sta_mask")
    n = int(g[f"{tag}_n"])
    ref = np.unpackbits(g[f"{tag}_maskbits"])[: n * n].reshape(n, n).astype(bool)
    t, te = (int(x) for x in g[f"{tag}_text"])
    m = O.sta_mask(_t(g[f"{tag}_latent"]), _t(g[f"{tag}_tile"]), _t(g[f"{tag}_window"]), t, te)
    assert np.array_equal(m, ref)


# ---------------------------------------------------------------- G4
@pytest.mark.parametrize("sp", [1, 2])
def test_tile_perm(golden, sp):
    g = golden("g4_tile_perm")
    assert np.array_equal(O.tile_major_order(LATENT, TILE, sp), g[f"sp{sp}_tiled_src"])


# ---------------------------------------------------------------- G5
def test_sliding_out(golden):
    g = golden("g5_sliding_out")
    o = O.sliding_tile_attention(g["wan_q"], g["wan_k"], g["wan_v"], LATENT, TILE, WINDOW)
    np.testing.assert_allclose(o, g["wan_out"], atol=2e-5, rtol=1e-4)
    t, te = (int(x) for x in g["text"])
    o, eo = O.sliding_tile_attention(g["hy_q"], g["hy_k"], g["hy_v"], LATENT, TILE, WINDOW,
                                     g["hy_eq"], g["hy_ek"], g["hy_ev"], te)
    np.testing.assert_allclose(o, g["hy_out"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(eo, g["hy_eout"], atol=2e-5, rtol=1e-4)
    assert np.all(eo[..., te:, :] == 0)  # padded text queries attend nothing


# ---------------------------------------------------------------- G6
def test_dense_out(golden):
    g = golden("g6_dense_out")
    t, te = (int(x) for x in g["hy_text"])
    o = O.dense_attention(g["hy_q"], g["hy_k"], g["hy_v"], kv_valid=S + te, q_valid=S + te)
    np.testing.assert_allclose(o[..., :S, :], g["hy_out"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(o[..., S:, :], g["hy_eout"], atol=2e-5, rtol=1e-4)
    assert np.all(g["hy_eout"][..., te:, :] == 0)
    np.testing.assert_allclose(O.dense_attention(g["wan_q"], g["wan_k"], g["wan_v"]), g["wan_out"], atol=2e-5,
                               rtol=1e-4)
    np.testing.assert_allclose(O.dense_attention(g["wan_q"], g["wan_kc"], g["wan_vc"]), g["wan_cross_out"],
                               atol=2e-5, rtol=1e-4)


# ---------------------------------------------------------------- G7
def test_router(golden):
    g = golden("g7_router")
    H = int(g["heads"])
    sc = O.router_scores(g["temb"], g["weight"], g["bias"], H)
    np.testing.assert_allclose(sc, g["scores"], atol=1e-6)
    for name, scores in (("router", g["scores"]), ("hand", g["hand_scores"])):
        for i, tau in enumerate(g["taus"]):
            e = O.route_heads(scores, float(tau))
            masks = np.stack([e == j for j in range(3)])
            assert np.array_equal(masks, g[f"{name}_head_masks"][i]), (name, tau)


# ---------------------------------------------------------------- G8
def _rmsnorm(x, w, eps=1e-6):
    return x / np.sqrt((x * x).mean(-1, keepdims=True) + eps) * w


def test_wan_eval_call(golden):
    g = golden("g8_eval_calls")
    H, D = 6, 16
    gi = O.group_info(LATENT, GROUP, 0.5)
    hidden = g["wan_hidden"].astype(np.float64)

    def lin(x, n):
        return x @ g[f"wan_w_{n}_weight"].astype(np.float64).T + g[f"wan_w_{n}_bias"]

    q = _rmsnorm(lin(hidden, "to_q"), g["wan_w_norm_q_weight"])
    k = _rmsnorm(lin(hidden, "to_k"), g["wan_w_norm_k_weight"])
    v = lin(hidden, "to_v")
    q, k, v = (a.reshape(1, S, H, D).transpose(0, 2, 1, 3) for a in (q, k, v))
    np.testing.assert_allclose(q, g["wan_q"], atol=2e-5)
    for tau in (0.3, 0.9):
        e = O.route_heads(g["routing_score"], tau)
        o = O.routed_attention(q, k, v, e, model="wan", latent=LATENT, tile=TILE, window=WINDOW, gi=gi)
        y = lin(o.transpose(0, 2, 1, 3).reshape(1, S, H * D), "to_out_0")
        np.testing.assert_allclose(y, g[f"wan_out_tau{int(tau*10)}"], atol=5e-5, rtol=1e-4)
    assert np.array_equal(O.route_heads(g["routing_score"], 0.9), np.zeros(6, dtype=np.int32))


def test_hunyuan_eval_steps(golden):
    g = golden("g8_eval_calls")
    gi = O.group_info(LATENT, GROUP, 0.5)
    t, te = (int(x) for x in g["text"])
    e = O.route_heads(g["routing_score"], 0.3)
    assert list(e) == [0, 1, 2, 0, 1, 2]
    o = O.routed_attention(g["hy_q"], g["hy_k"], g["hy_v"], e, model="hunyuan", latent=LATENT, tile=TILE,
                           window=WINDOW, gi=gi, t_text=t, t_eff=te)
    np.testing.assert_allclose(o[:, :, :S], g["hy_out"], atol=3e-5, rtol=1e-4)
    np.testing.assert_allclose(o[:, :, S:], g["hy_eout"], atol=3e-5, rtol=1e-4)
    assert np.all(g["hy_eout"][:, :, te:] == 0)


# ---------------------------------------------------------------- G11
def test_soft_mixture_forward(golden):
    """the training-time forward of the reference (all heads through all experts, score-weighted sum)"""
    g8, g = golden("g8_eval_calls"), golden("g11_soft_mixture")
    H, D = 6, 16
    gi = O.group_info(LATENT, GROUP, 0.5)
    sc = g["routing_score"]
    assert np.allclose(sc.sum(-1), 1.0, atol=1e-6)
    t, te = (int(x) for x in g8["text"])
    o = O.soft_mixture_attention(g8["hy_q"], g8["hy_k"], g8["hy_v"], sc, model="hunyuan", latent=LATENT, tile=TILE,
                                 window=WINDOW, gi=gi, t_text=t, t_eff=te)
    np.testing.assert_allclose(o[:, :, :S], g["hy_soft_out"], atol=3e-5, rtol=1e-4)
    np.testing.assert_allclose(o[:, :, S:], g["hy_soft_eout"], atol=3e-5, rtol=1e-4)

    def lin(x, n):
        return x @ g8[f"wan_w_{n}_weight"].astype(np.float64).T + g8[f"wan_w_{n}_bias"]

    hidden = g8["wan_hidden"].astype(np.float64)
    q = _rmsnorm(lin(hidden, "to_q"), g8["wan_w_norm_q_weight"])
    k = _rmsnorm(lin(hidden, "to_k"), g8["wan_w_norm_k_weight"])
    v = lin(hidden, "to_v")
    q, k, v = (a.reshape(1, S, H, D).transpose(0, 2, 1, 3) for a in (q, k, v))
    o = O.soft_mixture_attention(q, k, v, sc, model="wan", latent=LATENT, tile=TILE, window=WINDOW, gi=gi)
    y = lin(o.transpose(0, 2, 1, 3).reshape(1, S, H * D), "to_out_0")
    np.testing.assert_allclose(y, g["wan_soft_out"], atol=5e-5, rtol=1e-4)
    teacher = lin(O.dense_attention(q, k, v).transpose(0, 2, 1, 3).reshape(1, S, H * D), "to_out_0")
    np.testing.assert_allclose(teacher, g["wan_teacher_out"], atol=5e-5, rtol=1e-4)


# ---------------------------------------------------------------- G9
@pytest.mark.parametrize("P", [2, 4])
def test_ulysses_maps(golden, P):
    g = golden("g9_ulysses_maps")
    xs = [g[f"P{P}_r{r}_x"] for r in range(P)]
    ys = O.ulysses_seq_to_head(xs)
    for r in range(P):
        assert np.array_equal(ys[r], g[f"P{P}_r{r}_y"])
    zs = O.ulysses_head_to_seq(ys)
    for r in range(P):
        assert np.array_equal(zs[r], g[f"P{P}_r{r}_z"])
        assert np.array_equal(zs[r], xs[r])
        t = g[f"P{P}_r{r}_t"]
        assert np.array_equal(O.shrink_dim(t, 1, r, P), g[f"P{P}_r{r}_t_loc"])
    full = O.all_gather_cat([g[f"P{P}_r{r}_t_loc"] for r in range(P)], 1)
    assert np.array_equal(full, g[f"P{P}_r0_t_all"])


# ---------------------------------------------------------------- G10
def test_pixel2token(golden):
    g = golden("g10_pixel2token")
    for size, tok in zip(g["sizes"], g["tokens_hunyuan"]):
        assert O.video_to_latent(_t(size)) == _t(tok)
    assert np.array_equal(g["tokens_hunyuan"], g["tokens_wan"])
    for size, raises in zip(g["bad_sizes"], g["bad_raises"]):
        if raises:
            with pytest.raises(ValueError):
                O.video_to_latent(_t(size))
