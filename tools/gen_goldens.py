#!/usr/bin/env python
"""Generate golden vectors for the routed sparse-attention hot path by RUNNING THE REFERENCE.

Runs only in the authoring container (needs /root/reference, which never travels to the GPU box).
Output: small .npz fixtures under tests/golden/ (data only: inputs + expected outputs).

Harness notes:
  * `diffusers` is not installed here; the hot-path modules use it only for a type annotation
    (`Attention`) and for `apply_rotary_emb`, which sits BEFORE the attention boundary.  The harness
    registers empty stand-in modules so `import vorta.attention` succeeds; nothing from the
    stand-ins is ever executed for a fixture.
  * `torch.cuda.synchronize` is a no-op in the gloo workers (the reference calls it unconditionally
    after every all-to-all, vorta/ulysses/utils.py:49,81).
  * PYTHONDONTWRITEBYTECODE so nothing is written into /root/reference.

Usage:  python tools/gen_goldens.py [--only G1,G2,...]
"""
import argparse
import importlib.util
import os
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def import_reference():
    for name in ["diffusers", "diffusers.models", "diffusers.models.attention_processor",
                 "diffusers.models.embeddings"]:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["diffusers.models.attention_processor"].Attention = type("Attention", (), {})

    def _no_rope(*a, **k):
        raise RuntimeError("apply_rotary_emb is outside the attention boundary; not available in the harness")

    sys.modules["diffusers.models.embeddings"].apply_rotary_emb = _no_rope
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import vorta.attention as A  # noqa
    spec = importlib.util.spec_from_file_location("ref_router", os.path.join(REF, "vorta/patch/router.py"))
    router_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(router_mod)
    return A, router_mod


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    conv = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    np.savez_compressed(path, **conv)
    print(f"  wrote {path}  ({os.path.getsize(path)/1024:.1f} KiB)")


# ----------------------------------------------------------------------------- geometry used throughout
LATENT = (8, 6, 8)
TILE = (2, 3, 4)
WINDOW = (3, 3, 3)
GROUP = (2, 3, 2)
H, D = 6, 16
T_TXT, T_EFF = 16, 11
S = LATENT[0] * LATENT[1] * LATENT[2]


def g1_group_info(A):
    out = {}
    for tag, latent, win, r in [("4x6x4", (4, 6, 4), (2, 3, 2), 0.5),
                                ("8x6x8", (8, 6, 8), (2, 3, 2), 0.5),
                                ("9x6x8_g18", (9, 6, 8), (3, 3, 2), 0.5),
                                ("9x7x9_crop", (9, 7, 9), (2, 3, 2), 0.5),
                                ("8x6x8_r075", (8, 6, 8), (2, 3, 2), 0.75)]:
        gi = A.get_group_info(latent, win, reduction_rate=r)
        out[f"{tag}_latent"] = np.array(latent)
        out[f"{tag}_window"] = np.array(win)
        out[f"{tag}_rate"] = np.array(r)
        out[f"{tag}_center"] = gi.center_indices
        out[f"{tag}_margin"] = gi.margin_indices
        out[f"{tag}_num_unpooled"] = np.array(gi.num_unpooled_tokens_per_group)
    save("g1_group_info", **out)


def g2_pool_unpool(A):
    torch.manual_seed(1234)
    gi = A.get_group_info(LATENT, GROUP, reduction_rate=0.5)
    x = torch.randn(1, 2, S, D)
    y = torch.randn(1, 2, S, D)
    pooled, m = A.pool_sequence_by_similarity(x, gi, None)
    pooled_y, _ = A.pool_sequence_by_similarity(y, gi, m)  # V reusing K's matching
    unpooled = A.unpool_sequence_by_similarity(pooled, gi, m)
    # minimum similarity gap between rank n_u-1 and n_u (tie detector for index parity)
    save("g2_pool_unpool", latent=np.array(LATENT), group=np.array(GROUP), rate=np.array(0.5),
         x=x, y=y, pooled=pooled, pooled_y=pooled_y,
         unpooled_argsort=m.unpooled_argsort_sim, pooled_argsort=m.pooled_argsort_sim,
         unpooled=unpooled)


def _dense_mask(mask_mod, n):
    q = torch.arange(n).view(n, 1).expand(n, n)
    kv = torch.arange(n).view(1, n).expand(n, n)
    z = torch.zeros((), dtype=torch.long)
    return mask_mod(z, z, q, kv)


def _ref_mask_mod(A, **kw):
    """Evaluate the reference's mask_mod closure without building a BlockMask (capture it)."""
    mod = A.sliding_attn_flex
    orig = mod.create_block_mask
    captured = {}

    def grab(fn, **kwargs):
        captured["fn"] = fn
        captured["kw"] = kwargs
        return fn

    mod.create_block_mask = grab
    try:
        mod.create_sliding_tile_attn_mask_func(device=torch.device("cpu"), **kw)
    finally:
        mod.create_block_mask = orig
    return captured["fn"], captured["kw"]


def g3_sta_mask(A):
    out = {}
    cases = {
        "hy_text": dict(latent_shape=LATENT, window_size=WINDOW, tile_size=TILE, text_seq_length=T_TXT,
                        text_seq_length_no_pad=T_EFF),
        "wan_notext": dict(latent_shape=LATENT, window_size=WINDOW, tile_size=TILE, text_seq_length=0,
                           text_seq_length_no_pad=0),
        "narrow_t": dict(latent_shape=(4, 6, 8), window_size=WINDOW, tile_size=TILE, text_seq_length=8,
                         text_seq_length_no_pad=5),
        "win531": dict(latent_shape=(10, 6, 8), window_size=(5, 3, 1), tile_size=(2, 3, 4), text_seq_length=0,
                       text_seq_length_no_pad=0),
    }
    for tag, kw in cases.items():
        fn, bm_kw = _ref_mask_mod(A, **kw)
        n = bm_kw["Q_LEN"]
        m = _dense_mask(fn, n).numpy().astype(np.uint8)
        out[f"{tag}_latent"] = np.array(kw["latent_shape"])
        out[f"{tag}_window"] = np.array(kw["window_size"])
        out[f"{tag}_tile"] = np.array(kw["tile_size"])
        out[f"{tag}_text"] = np.array([kw["text_seq_length"], kw["text_seq_length_no_pad"]])
        out[f"{tag}_n"] = np.array(n)
        out[f"{tag}_maskbits"] = np.packbits(m, axis=None)
    save("g3_sta_mask", **out)


def g4_tile_perm(A):
    out = {}
    for sp in (1, 2):
        idx = torch.arange(S, dtype=torch.float32).view(1, 1, S, 1)
        tiled = A.tile.tile_layout(idx, sp_size=sp, tile_size=TILE, latent_shape=LATENT, head_dim=1)
        back = A.tile.untile_layout(tiled, sp_size=sp, tile_size=TILE, latent_shape=LATENT, head_dim=1)
        assert torch.equal(back, idx)
        out[f"sp{sp}_tiled_src"] = tiled.view(-1).to(torch.int64)  # tiled[i] = source raster index
    out["latent"] = np.array(LATENT)
    out["tile"] = np.array(TILE)
    save("g4_tile_perm", **out)


def _flex(A):
    """The reference's compiled flex_attention; fall back to eager flex if Inductor fails here."""
    from torch.nn.attention.flex_attention import flex_attention as eager_flex
    return A.hunyuan.flex_attention, eager_flex


def _block_mask(A, **kw):
    from torch.nn.attention.flex_attention import create_block_mask
    fn, bm_kw = _ref_mask_mod(A, **kw)
    return create_block_mask(fn, B=None, H=None, Q_LEN=bm_kw["Q_LEN"], KV_LEN=bm_kw["KV_LEN"],
                             device="cpu", _compile=False)


def _run_flex(A, q, k, v, bm):
    compiled, eager = _flex(A)
    try:
        return compiled(q, k, v, block_mask=bm), "compiled"
    except Exception as e:  # pragma: no cover
        print("   compiled flex_attention failed here (", type(e).__name__, ") -> eager flex_attention")
        return eager(q, k, v, block_mask=bm), "eager"


def g5_sliding_out(A):
    torch.manual_seed(2345)
    out = {"latent": np.array(LATENT), "tile": np.array(TILE), "window": np.array(WINDOW),
           "text": np.array([T_TXT, T_EFF])}
    # Wan: text free
    q, k, v = (torch.randn(1, 2, S, D) for _ in range(3))
    bm = _block_mask(A, latent_shape=LATENT, window_size=WINDOW, tile_size=TILE, text_seq_length=0,
                     text_seq_length_no_pad=0)
    mode = {}

    def flex_fn(q_, k_, v_):
        o, mode["m"] = _run_flex(A, q_, k_, v_, bm)
        return o

    o = A.sliding_attn_flex.sliding_tile_flex_attn(q, k, v, flex_fn, tile_size=TILE, latent_shape=LATENT, head_dim=1)
    out.update(wan_q=q, wan_k=k, wan_v=v, wan_out=o)
    # Hunyuan: with (padded) text
    q, k, v = (torch.randn(1, 2, S, D) for _ in range(3))
    eq, ek, ev = (torch.randn(1, 2, T_TXT, D) for _ in range(3))
    bm2 = _block_mask(A, latent_shape=LATENT, window_size=WINDOW, tile_size=TILE, text_seq_length=T_TXT,
                      text_seq_length_no_pad=T_EFF)

    def flex_fn2(q_, k_, v_):
        o, mode["m"] = _run_flex(A, q_, k_, v_, bm2)
        return o

    o, eo = A.sliding_attn_flex.sliding_tile_flex_attn(q, k, v, flex_fn2, eq, ek, ev, tile_size=TILE,
                                                       latent_shape=LATENT, head_dim=1)
    out.update(hy_q=q, hy_k=k, hy_v=v, hy_eq=eq, hy_ek=ek, hy_ev=ev, hy_out=o, hy_eout=eo)
    out["flex_mode"] = np.array(mode["m"])
    save("g5_sliding_out", **out)


def _hy_mask(n_video, t, t_eff):
    m = torch.zeros(1, 1, 1, n_video + t, dtype=torch.bool)
    m[..., : n_video + t_eff] = True
    return m


def g6_dense_out(A):
    torch.manual_seed(3456)
    out = {}
    proc = A.HunyuanVideoFlashAttnProcessor()
    q, k, v = (torch.randn(1, 3, S + T_TXT, D) for _ in range(3))
    mask = _hy_mask(S, T_TXT, T_EFF)
    o, eo = proc._step_attention(q, k, v, mask, T_TXT)
    out.update(hy_q=q, hy_k=k, hy_v=v, hy_text=np.array([T_TXT, T_EFF]), hy_out=o, hy_eout=eo)
    wproc = A.WanAttnProcessor2_0()
    q = torch.randn(1, 3, S, D)
    k, v = (torch.randn(1, 3, S, D) for _ in range(2))
    o, _ = wproc._attn(None, q, k, v, None, None, is_cross_attn=False)
    out.update(wan_q=q, wan_k=k, wan_v=v, wan_out=o)
    kc, vc = (torch.randn(1, 3, 40, D) for _ in range(2))  # cross: Sq != Skv
    o, _ = wproc._attn(None, q, kc, vc, None, None, is_cross_attn=True)
    out.update(wan_kc=kc, wan_vc=vc, wan_cross_out=o)
    save("g6_dense_out", **out)


def g7_router(A, router_mod):
    torch.manual_seed(4567)
    e_dim = 48
    r = router_mod.Router(e_dim, H)
    temb = torch.randn(2, e_dim)
    with torch.no_grad():
        scores = r(temb)
    out = dict(weight=r.linear.weight, bias=r.linear.bias, temb=temb, scores=scores, heads=np.array(H))
    proc = A.HunyuanVideoFlashAttnProcessorTripleEval()
    wproc = A.WanAttnProcessorTripleEval()
    q = torch.zeros(2, H, 4, 2)
    # a hand-made score table that exercises the threshold: max in {0.34, 0.4, 0.5, 0.6, 0.9, 1/3}
    hand = torch.tensor([[[0.30, 0.34, 0.36], [0.40, 0.35, 0.25], [0.25, 0.25, 0.50],
                          [0.20, 0.60, 0.20], [0.05, 0.05, 0.90], [1 / 3, 1 / 3, 1 / 3]]])
    hand = torch.cat([hand, hand.flip(-1)], dim=0)  # batch item 1 differs: must be ignored
    taus = [0.0, 0.3, 0.34, 0.5, 0.9]
    for name, sc in (("router", scores), ("hand", hand)):
        masks = []
        for tau in taus:
            res = proc._get_routed_qkv(q, q, q, sc.clone(), tau)
            res_w = wproc._get_routed_qkv(q, q, q, sc.clone(), tau)
            m = torch.stack([res[3], res[7], res[11]])
            mw = torch.stack([res_w[3], res_w[7], res_w[11]])
            assert torch.equal(m, mw)
            masks.append(m)
        out[f"{name}_head_masks"] = torch.stack(masks)  # (n_tau, 3, H) bool
    out["hand_scores"] = hand
    out["taus"] = np.array(taus)
    save("g7_router", **out)


class _FakeAttn(torch.nn.Module):
    """Build-owned stand-in for diffusers' Attention: only the attributes the processors touch."""

    def __init__(self, heads, dim_head, wan):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.to_q = torch.nn.Linear(inner, inner)
        self.to_k = torch.nn.Linear(inner, inner)
        self.to_v = torch.nn.Linear(inner, inner)
        if wan:
            self.norm_q = torch.nn.RMSNorm(inner, eps=1e-6)
            self.norm_k = torch.nn.RMSNorm(inner, eps=1e-6)
        else:
            self.norm_q = torch.nn.RMSNorm(dim_head, eps=1e-6)
            self.norm_k = torch.nn.RMSNorm(dim_head, eps=1e-6)
        self.add_q_proj = None
        self.add_k_proj = None
        self.add_v_proj = None
        self.to_out = torch.nn.ModuleList([torch.nn.Linear(inner, inner), torch.nn.Identity()])
        self.to_add_out = None


def g8_eval_calls(A):
    out = {"latent": np.array(LATENT), "tile": np.array(TILE), "window": np.array(WINDOW),
           "group": np.array(GROUP), "rate": np.array(0.5), "text": np.array([T_TXT, T_EFF])}
    gi = A.get_group_info(LATENT, GROUP, reduction_rate=0.5)
    # routing scores: heads 0,3 -> full ; 1,4 -> lowres ; 2,5 -> sliding  (top-1 well above tau)
    sc = torch.full((1, H, 3), 0.1)
    for h in range(H):
        sc[0, h, h % 3] = 0.8
    out["routing_score"] = sc

    # ---- Wan: the whole processor call (self-attention, no RoPE tensor: RoPE sits before the boundary)
    torch.manual_seed(5678)
    attn = _FakeAttn(H, D, wan=True)
    with torch.no_grad():
        attn.norm_q.weight.uniform_(0.5, 1.5)
        attn.norm_k.weight.uniform_(0.5, 1.5)
    hidden = torch.randn(1, S, H * D)
    bm = _block_mask(A, latent_shape=LATENT, window_size=WINDOW, tile_size=TILE, text_seq_length=0,
                     text_seq_length_no_pad=0)
    wproc = A.WanAttnProcessorTripleEval(check_input=True)
    for tau in (0.3, 0.9):
        with torch.no_grad():
            y = wproc(attn, hidden, None, None, None, tau_sparse=tau, routing_score=sc.clone(), lowres_group_info=gi,
                      flex_attn_mask_func=bm, window_size=WINDOW, tile_size=TILE, latent_shape=LATENT)
        out[f"wan_out_tau{int(tau*10)}"] = y
    with torch.no_grad():
        q, k, v, _ = wproc._input_proj(attn, hidden, None, None)
    out.update(wan_hidden=hidden, wan_q=q, wan_k=k, wan_v=v)
    for n, p in attn.state_dict().items():
        out["wan_w_" + n.replace(".", "_")] = p

    # ---- Hunyuan: the expert steps + combine on post-RoPE q,k,v (with padded text)
    torch.manual_seed(6789)
    q, k, v = (torch.randn(1, H, S + T_TXT, D) for _ in range(3))
    mask = _hy_mask(S, T_TXT, T_EFF)
    bm2 = _block_mask(A, latent_shape=LATENT, window_size=WINDOW, tile_size=TILE, text_seq_length=T_TXT,
                      text_seq_length_no_pad=T_EFF)
    proc = A.HunyuanVideoFlashAttnProcessorTripleEval()
    with torch.no_grad():
        (fq, fk, fv, fm, lq, lk, lv, lm, sq, sk, sv, sm) = proc._get_routed_qkv(q, k, v, sc.clone(), 0.3)
        fo, feo = proc._step_attention(fq, fk, fv, mask, T_TXT)
        lo, leo = proc._step_lowres_attention(lq, lk, lv, mask, T_TXT, gi)
        so, seo = proc._step_sliding_attention(sq, sk, sv, T_TXT, bm2, TILE, LATENT)
        o = proc._combine_attn_outputs(H, [fo, lo, so], [fm, lm, sm])
        eo = proc._combine_attn_outputs(H, [feo, leo, seo], [fm, lm, sm])
    out.update(hy_q=q, hy_k=k, hy_v=v, hy_out=o, hy_eout=eo,
               hy_lowres_only=lo, hy_lowres_eonly=leo)
    save("g8_eval_calls", **out)


# ----------------------------------------------------------------------------- G11: soft-mixture (training) forward
def g11_soft_mixture(A):
    """The training-time forward (soft mixture of the three experts over ALL heads): wan.py:195-241,296-300 (whole
    processor call) and hunyuan.py:341-408,509-513 (expert steps + combine).  Inputs are G8's (same seeds), so only
    the soft scores and the outputs are stored."""
    g8 = np.load(os.path.join(OUT, "g8_eval_calls.npz"))
    gi = A.get_group_info(LATENT, GROUP, reduction_rate=0.5)
    torch.manual_seed(4242)
    sc = torch.softmax(torch.randn(1, H, 3) * 1.5, dim=-1)
    out = {"routing_score": sc}
    torch.manual_seed(5678)
    attn = _FakeAttn(H, D, wan=True)
    with torch.no_grad():
        attn.norm_q.weight.uniform_(0.5, 1.5)
        attn.norm_k.weight.uniform_(0.5, 1.5)
    hidden = torch.randn(1, S, H * D)
    assert np.array_equal(hidden.numpy(), g8["wan_hidden"]) and \
        np.array_equal(attn.to_q.weight.detach().numpy(), g8["wan_w_to_q_weight"])
    bm = _block_mask(A, latent_shape=LATENT, window_size=WINDOW, tile_size=TILE, text_seq_length=0,
                     text_seq_length_no_pad=0)
    wproc = A.WanAttnProcessorTripleTrain(check_input=True)
    with torch.no_grad():
        y = wproc(attn, hidden, None, None, None, routing_score=sc.clone(), lowres_group_info=gi,
                  flex_attn_mask_func=bm, window_size=WINDOW, tile_size=TILE, latent_shape=LATENT)
        teacher = wproc(attn, hidden, None, None, None, use_original_attn=True)
    out.update(wan_soft_out=y, wan_teacher_out=teacher)

    q, k, v = (torch.tensor(g8[n]) for n in ("hy_q", "hy_k", "hy_v"))
    mask = _hy_mask(S, T_TXT, T_EFF)
    bm2 = _block_mask(A, latent_shape=LATENT, window_size=WINDOW, tile_size=TILE, text_seq_length=T_TXT,
                      text_seq_length_no_pad=T_EFF)
    proc = A.HunyuanVideoFlashAttnProcessorTripleTrain()
    with torch.no_grad():
        fo, feo = proc._step_attention(q, k, v, mask, T_TXT)
        lo, leo = proc._step_lowres_attention(q, k, v, mask, T_TXT, gi)
        so, seo = proc._step_sliding_attention(q, k, v, T_TXT, bm2, TILE, LATENT)
        o = proc._combine_attn_outputs(sc, [fo, lo, so])
        eo = proc._combine_attn_outputs(sc, [feo, leo, seo])
    out.update(hy_soft_out=o, hy_soft_eout=eo)
    save("g11_soft_mixture", **out)


# ----------------------------------------------------------------------------- G9: Ulysses maps under gloo
def _g9_worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.cuda.synchronize = lambda *a, **k: None
    import_reference()
    from vorta.ulysses import SP_STATE, all_gather, all_to_all_4D, shrink_dim
    dist.init_process_group("gloo", rank=rank, world_size=world)
    SP_STATE.setup_sp_group(world)
    Hh, Sl, Dd = 8, 6, 2
    # tag = global_head * 1000 + global_seq (+ d*0.5): sequence-sharded input (B,H,S/P,D)
    h = torch.arange(Hh).view(1, Hh, 1, 1)
    s = (torch.arange(Sl) + rank * Sl).view(1, 1, Sl, 1)
    d = torch.arange(Dd).view(1, 1, 1, Dd)
    x = (h * 1000 + s).float() + d * 0.25
    y = all_to_all_4D(x, scatter_idx=1, gather_idx=2)  # (B,H/P,S,D)
    z = all_to_all_4D(y, scatter_idx=2, gather_idx=1)  # back
    t = (torch.arange(Hh).view(1, Hh, 1, 1) * 10 + torch.arange(3).view(1, 1, 3, 1)).float().expand(1, Hh, 3, Dd)
    t_loc = shrink_dim(t, dim=1).contiguous()
    t_all = all_gather(t_loc, dim=1)
    ret[rank] = dict(x=x.numpy(), y=y.numpy(), z=z.numpy(), t=t.numpy().copy(), t_loc=t_loc.numpy(),
                     t_all=t_all.numpy())
    dist.barrier()
    dist.destroy_process_group()


def g9_ulysses():
    import torch.multiprocessing as mp
    out = {}
    for world, port in ((2, 29611), (4, 29612), (8, 29613)):
        mgr = mp.Manager()
        ret = mgr.dict()
        mp.spawn(_g9_worker, args=(world, port, ret), nprocs=world, join=True)
        for r in range(world):
            for k, v in ret[r].items():
                out[f"P{world}_r{r}_{k}"] = v
    save("g9_ulysses_maps", **out)


def g10_pixel2token():
    # vorta/patch/__init__.py is empty, and utils.py only needs `..attention`
    import_reference()
    U = importlib.import_module("vorta.patch.utils")
    sizes = [(49, 320, 512), (81, 480, 832), (129, 720, 1280), (81, 720, 1280), (117, 720, 1280), (77, 720, 1280),
             (77, 480, 832), (117, 768, 1280)]
    res = np.array([U.hunyuan_pixel2token(s) for s in sizes])
    res_w = np.array([U.wan_pixel2token(s) for s in sizes])
    bad = []
    for s in [(50, 320, 512), (49, 322, 512), (49, 320, 515)]:
        try:
            U.hunyuan_pixel2token(s)
            bad.append(0)
        except ValueError:
            bad.append(1)
    save("g10_pixel2token", sizes=np.array(sizes), tokens_hunyuan=res, tokens_wan=res_w,
         bad_sizes=np.array([(50, 320, 512), (49, 322, 512), (49, 320, 515)]), bad_raises=np.array(bad))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    only = set(filter(None, args.only.split(",")))
    A, router_mod = import_reference()
    torch.set_num_threads(4)
    jobs = [("G1", lambda: g1_group_info(A)), ("G2", lambda: g2_pool_unpool(A)), ("G3", lambda: g3_sta_mask(A)),
            ("G4", lambda: g4_tile_perm(A)), ("G5", lambda: g5_sliding_out(A)), ("G6", lambda: g6_dense_out(A)),
            ("G7", lambda: g7_router(A, router_mod)), ("G8", lambda: g8_eval_calls(A)), ("G9", g9_ulysses),
            ("G10", g10_pixel2token), ("G11", lambda: g11_soft_mixture(A))]
    for name, fn in jobs:
        if only and name not in only:
            continue
        print(name)
        fn()


if __name__ == "__main__":
    main()
