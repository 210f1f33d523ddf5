"""CPU ORACLE for the routed sparse-attention denoising path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain numpy restatement of the reference algorithm (wenhao728/VORTA) for the hot path named in
BASELINE.json.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module; the product path (`vorta_amd/`) never does and fails loudly without its HIP library.

Parity status: PINNED.  Every function below is checked against golden vectors produced by running the
reference itself in the authoring container (tools/gen_goldens.py -> tests/golden/*.npz; see
tests/test_oracle_golden.py).

All citations are relative to /root/reference/.  Arithmetic is float64 unless `dtype` says otherwise,
so that the oracle is a stricter target than the reference's own bf16/fp32 execution.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

Triple = Tuple[int, int, int]


# ----------------------------------------------------------------------------------------------- A4
@dataclass
class GroupInfo:
    """vorta/attention/coreset_select.py:8-12 (LowresGroupInfo)."""
    center: np.ndarray  # (G, 1) int64
    margin: np.ndarray  # (G, g-1) int64
    n_keep_margin: int  # "num_unpooled_tokens_per_group"

    @property
    def n_groups(self) -> int:
        return self.center.shape[0]

    @property
    def group_size(self) -> int:
        return 1 + self.margin.shape[1]


def group_info(latent: Triple, window: Triple, rate: float = 0.5) -> GroupInfo:
    """vorta/attention/coreset_select.py:15-60.

    Raster index cube cropped to whole windows, regrouped so each row lists one window's tokens in
    in-window raster order; the centre is the token at (fw//2, hw//2, ww//2)."""
    f, h, w = latent
    fw, hw, ww = window
    fg, hg, wg = f // fw, h // hw, w // ww
    cube = np.arange(f * h * w, dtype=np.int64).reshape(f, h, w)[: fg * fw, : hg * hw, : wg * ww]
    groups = cube.reshape(fg, fw, hg, hw, wg, ww).transpose(0, 2, 4, 1, 3, 5).reshape(fg * hg * wg, fw * hw * ww)
    c = (fw // 2) * hw * ww + (hw // 2) * ww + ww // 2
    center = groups[:, c:c + 1]
    margin = np.concatenate([groups[:, :c], groups[:, c + 1:]], axis=1)
    n_keep = int(fw * hw * ww * (1 - rate)) - 1
    return GroupInfo(center=center, margin=margin, n_keep_margin=n_keep)


# ----------------------------------------------------------------------------------------------- A5 / A7
def _l2_normalize(x: np.ndarray, eps: float = 1e-12) -> np.ndarray:
    # torch.nn.functional.normalize: x / max(||x||_2, eps)   (coreset_select.py:100-101)
    n = np.sqrt((x * x).sum(-1, keepdims=True))
    return x / np.maximum(n, eps)


def coreset_similarity(x: np.ndarray, gi: GroupInfo) -> np.ndarray:
    """cos-sim(centre, each margin) -> (B,h,G,g-1).  coreset_select.py:91-103."""
    c = x[:, :, gi.center[:, 0], :]  # (B,h,G,D)
    m = x[:, :, gi.margin, :]  # (B,h,G,g-1,D)
    return np.einsum("bhgd,bhgmd->bhgm", _l2_normalize(c), _l2_normalize(m))


def coreset_match(x: np.ndarray, gi: GroupInfo) -> Tuple[np.ndarray, np.ndarray]:
    """Ascending argsort of the similarities, split into (kept, dropped) margin slots.

    coreset_select.py:105-114.  Ties are broken by slot index (stable sort); the reference's argsort is
    unstable, so index parity is only asserted on tie-free data."""
    order = np.argsort(coreset_similarity(x, gi), axis=-1, kind="stable")
    return order[..., : gi.n_keep_margin], order[..., gi.n_keep_margin:]


def coreset_pool(x: np.ndarray, gi: GroupInfo, kept: np.ndarray) -> np.ndarray:
    """Packed sequence [G centres | G x n_keep kept margins (group-major, least similar first)].
    coreset_select.py:116-124."""
    B, h, _, D = x.shape
    c = x[:, :, gi.center[:, 0], :]
    m = x[:, :, gi.margin, :]
    km = np.take_along_axis(m, kept[..., None], axis=3)  # (B,h,G,n_keep,D)
    return np.concatenate([c, km.reshape(B, h, -1, D)], axis=2)


def coreset_row_lists(gi: GroupInfo, kept: np.ndarray, dropped: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Raster token ids of the packed sequence, and per-group raster ids of the dropped margins.

    keep_rows (B,h,G*(1+n_keep)); drop_rows (B,h,G,g-1-n_keep).  This is the index form of
    coreset_select.py:116-124 (gather) and :159-166 (scatter destinations)."""
    B, h = kept.shape[:2]
    margin = np.broadcast_to(gi.margin[None, None], (B, h) + gi.margin.shape)
    kept_rows = np.take_along_axis(margin, kept, axis=3).reshape(B, h, -1)
    drop_rows = np.take_along_axis(margin, dropped, axis=3)
    centres = np.broadcast_to(gi.center[:, 0][None, None], (B, h, gi.n_groups))
    return np.concatenate([centres, kept_rows], axis=2), drop_rows


def coreset_unpool(y: np.ndarray, gi: GroupInfo, kept: np.ndarray, dropped: np.ndarray) -> np.ndarray:
    """Scatter packed rows back; every dropped margin receives its centre's row; tokens outside any
    whole window stay zero.  coreset_select.py:127-185."""
    B, h, _, D = y.shape
    G, g = gi.n_groups, gi.group_size
    out = np.zeros((B, h, G * g, D), dtype=y.dtype)
    keep_rows, drop_rows = coreset_row_lists(gi, kept, dropped)
    centres = y[:, :, :G, :]
    for b in range(B):
        for hh in range(h):
            out[b, hh, keep_rows[b, hh], :] = y[b, hh]
            out[b, hh, drop_rows[b, hh].reshape(-1), :] = np.repeat(centres[b, hh], drop_rows.shape[-1], axis=0)
    return out


# ----------------------------------------------------------------------------------------------- A9
def tile_major_order(latent: Triple, tile: Triple, sp: int = 1) -> np.ndarray:
    """perm[i] = raster index of the token at tile-major position i.  vorta/attention/tile.py:7-41.

    For sp>1 the reference first regroups '(sp t h w) -> (t sp h w)' (tile.py:21-25), which interleaves
    frames; restated here so the quirk is documented by a fixture (g4)."""
    t, h, w = latent
    tt, th, tw = tile
    src = np.arange(t * h * w, dtype=np.int64)
    if sp > 1:
        src = src.reshape(sp, t // sp, h, w).transpose(1, 0, 2, 3).reshape(-1)
    nt, nh, nw = t // tt, h // th, w // tw
    return src.reshape(nt, tt, nh, th, nw, tw).transpose(0, 2, 4, 1, 3, 5).reshape(-1)


# ----------------------------------------------------------------------------------------------- A8
def sta_window_tiles(latent: Triple, tile: Triple, window: Triple) -> np.ndarray:
    """Boolean (n_tiles, n_tiles): may q-tile i see kv-tile j?  sliding_attn_flex.py:93-127.

    Per dim the window centre is clamp(q_tile, w//2, n-1-w//2) with torch.clamp semantics
    (min applied first, then max: when min>max the result is max)."""
    n = [latent[i] // tile[i] for i in range(3)]
    coords = np.stack(np.meshgrid(*[np.arange(x) for x in n], indexing="ij"), -1).reshape(-1, 3)
    ok = np.ones((coords.shape[0], coords.shape[0]), dtype=bool)
    for d in range(3):
        half = window[d] // 2
        centre = np.minimum(np.maximum(coords[:, d], half), (n[d] - 1) - half)
        ok &= np.abs(centre[:, None] - coords[None, :, d]) <= half
    return ok


def sta_mask(latent: Triple, tile: Triple, window: Triple, t_text: int, t_eff: int) -> np.ndarray:
    """Dense boolean (S+T, S+T) mask in TILE-MAJOR token order.  sliding_attn_flex.py:101-128."""
    S = latent[0] * latent[1] * latent[2]
    tok = tile[0] * tile[1] * tile[2]
    n = S + t_text
    m = np.zeros((n, n), dtype=bool)
    tiles = sta_window_tiles(latent, tile, window)
    tid = np.arange(S) // tok
    m[:S, :S] = tiles[tid][:, tid]
    m[:S, S:S + t_eff] = True  # video -> valid text
    m[S:S + t_eff, :S + t_eff] = True  # valid text -> everything valid
    return m


# ----------------------------------------------------------------------------------------------- A3
def _softmax_attend(q: np.ndarray, k: np.ndarray, v: np.ndarray, mask: Optional[np.ndarray] = None,
                    scale: Optional[float] = None, dtype=np.float64) -> np.ndarray:
    """softmax(q k^T * scale [masked]) v over the last two dims; fully-masked rows -> 0.
    (`dtype=np.float32` is used only by bench.py's cpu_baseline leg, to time the port at a fair precision.)"""
    q, k, v = (np.asarray(a, dtype=dtype) for a in (q, k, v))
    scale = 1.0 / np.sqrt(q.shape[-1]) if scale is None else scale
    s = np.matmul(q, np.swapaxes(k, -1, -2)) * scale  # BLAS-backed (einsum is not)
    if mask is not None:
        s = np.where(mask, s, -np.inf)
    mx = s.max(-1, keepdims=True)
    mx = np.where(np.isfinite(mx), mx, 0.0)
    p = np.exp(s - mx)
    den = p.sum(-1, keepdims=True)
    p = np.divide(p, den, out=np.zeros_like(p), where=den > 0)
    return np.matmul(p, v)


def dense_attention(q, k, v, kv_valid: Optional[int] = None, q_valid: Optional[int] = None) -> np.ndarray:
    """Full attention on the first `kv_valid` keys; query rows >= q_valid are zero.

    Hunyuan: hunyuan.py:167-176 (L = mask.sum(); SDPA on [:L]; zero-pad the rest).
    Wan:     wan.py:142-145 (no mask), incl. cross attention with Sq != Skv."""
    Sq, Skv = q.shape[-2], k.shape[-2]
    kv_valid = Skv if kv_valid is None else kv_valid
    q_valid = Sq if q_valid is None else q_valid
    out = np.zeros(q.shape[:-1] + (v.shape[-1],), dtype=np.float64)
    out[..., :q_valid, :] = _softmax_attend(q[..., :q_valid, :], k[..., :kv_valid, :], v[..., :kv_valid, :])
    return out


# ----------------------------------------------------------------------------------------------- A10
def sliding_tile_attention(q, k, v, latent: Triple, tile: Triple, window: Triple,
                           eq=None, ek=None, ev=None, t_eff: int = 0):
    """Sliding-tile expert on RASTER-ordered q,k,v (B,h,S,D) [+ text (B,h,T,D)].

    Restates tile -> concat text -> masked attention -> split -> untile
    (sliding_attn_flex.py:137-211 with the mask of :101-128).  Returns video out (raster order) and, if
    text is given, the text out."""
    S = q.shape[-2]
    perm = tile_major_order(latent, tile)
    has_text = eq is not None
    T = eq.shape[-2] if has_text else 0
    qq, kk, vv = q[..., perm, :], k[..., perm, :], v[..., perm, :]
    if has_text:
        qq = np.concatenate([qq, eq], -2)
        kk = np.concatenate([kk, ek], -2)
        vv = np.concatenate([vv, ev], -2)
    m = sta_mask(latent, tile, window, T, t_eff)
    o = _softmax_attend(qq, kk, vv, mask=m)
    out = np.empty_like(o[..., :S, :])
    out[..., perm, :] = o[..., :S, :]
    return (out, o[..., S:, :]) if has_text else out


# ----------------------------------------------------------------------------------------------- A6
def lowres_attention(q, k, v, gi: GroupInfo, model: str, eq=None, ek=None, ev=None, t_eff: int = 0,
                     matches=None):
    """Coreset ("low-res") expert.

    hunyuan (hunyuan.py:410-457): Q and K matched independently, V reuses K's; text appended after the
      packed video tokens; keys limited to S_low + t_eff; padded text query rows are zero.
    wan (wan.py:243-270): K and V reuse Q's matching; no text.
    `matches` optionally supplies ((q_kept,q_drop),(k_kept,k_drop)) to make index choices external."""
    if matches is None:
        mq = coreset_match(q, gi)
        mk = coreset_match(k, gi) if model == "hunyuan" else mq
    else:
        mq, mk = matches
    ql = coreset_pool(q, gi, mq[0])
    kl = coreset_pool(k, gi, mk[0])
    vl = coreset_pool(v, gi, mk[0])
    s_low = ql.shape[-2]
    if eq is not None:
        T = eq.shape[-2]
        ql, kl, vl = (np.concatenate([a, b], -2) for a, b in ((ql, eq), (kl, ek), (vl, ev)))
        o = dense_attention(ql, kl, vl, kv_valid=s_low + t_eff, q_valid=s_low + t_eff)
        return coreset_unpool(o[..., :s_low, :], gi, *mq), o[..., s_low:s_low + T, :]
    o = dense_attention(ql, kl, vl)
    return coreset_unpool(o, gi, *mq)


# ----------------------------------------------------------------------------------------------- A1 / A2
def router_scores(temb: np.ndarray, weight: np.ndarray, bias: np.ndarray, heads: int) -> np.ndarray:
    """softmax((W silu(temb) + b).view(B,H,3)).  vorta/patch/router.py:33-43."""
    x = np.asarray(temb, dtype=np.float64)
    x = x / (1.0 + np.exp(-x))
    y = (x @ np.asarray(weight, dtype=np.float64).T + np.asarray(bias, dtype=np.float64)).reshape(x.shape[0], heads, -1)
    y = y - y.max(-1, keepdims=True)
    e = np.exp(y)
    return e / e.sum(-1, keepdims=True)


def route_heads(scores: np.ndarray, tau: float) -> np.ndarray:
    """expert id per head from batch item 0: top-1, falling back to expert 0 when its score < tau.
    hunyuan.py:620-624 == wan.py:396-400.  (torch.topk returns the FIRST maximal index on ties.)"""
    s = np.asarray(scores)[0]
    idx = s.argmax(-1)
    top = s.max(-1)
    return np.where(top < tau, 0, idx).astype(np.int32)


# ----------------------------------------------------------------------------------------------- A11 / A12
def routed_attention(q, k, v, expert_of_head: np.ndarray, *, model: str, latent: Triple, tile: Triple,
                     window: Triple, gi: GroupInfo, t_text: int = 0, t_eff: int = 0):
    """The whole routed op on post-RoPE q,k,v (B,H,S+T,D): dispatch by head, three experts, combine.

    hunyuan.py:556-605 / wan.py:351-383.  Returns (B,H,S+T,D) with text rows at the end (hunyuan) or
    (B,H,S,D) (wan)."""
    B, H, N, D = q.shape
    S = latent[0] * latent[1] * latent[2]
    out = np.zeros((B, H, N, D), dtype=np.float64)
    hy = model == "hunyuan"
    for e in range(3):
        hs = np.nonzero(expert_of_head == e)[0]
        if hs.size == 0:
            continue
        qe, ke, ve = q[:, hs], k[:, hs], v[:, hs]
        if e == 0:
            out[:, hs] = dense_attention(qe, ke, ve, kv_valid=S + t_eff, q_valid=S + t_eff) if hy else \
                dense_attention(qe, ke, ve)
        elif e == 1:
            if hy:
                o, eo = lowres_attention(qe[..., :S, :], ke[..., :S, :], ve[..., :S, :], gi, "hunyuan",
                                         qe[..., S:, :], ke[..., S:, :], ve[..., S:, :], t_eff)
                out[:, hs, :S], out[:, hs, S:] = o, eo
            else:
                out[:, hs] = lowres_attention(qe, ke, ve, gi, "wan")
        else:
            if hy:
                o, eo = sliding_tile_attention(qe[..., :S, :], ke[..., :S, :], ve[..., :S, :], latent, tile, window,
                                               qe[..., S:, :], ke[..., S:, :], ve[..., S:, :], t_eff)
                out[:, hs, :S], out[:, hs, S:] = o, eo
            else:
                out[:, hs] = sliding_tile_attention(qe, ke, ve, latent, tile, window)
    return out


def soft_mixture_attention(q, k, v, scores: np.ndarray, *, model: str, latent: Triple, tile: Triple, window: Triple,
                           gi: GroupInfo, t_text: int = 0, t_eff: int = 0):
    """Training-time forward: every head runs all three experts and the outputs are mixed with the routing
    scores, out[b,h] = sum_e scores[b,h,e] * expert_e(q,k,v)[b,h].
    hunyuan.py:341-408 + `_combine_attn_outputs` :509-513;  wan.py:218-241 + :296-300.  """
    H = q.shape[1]
    sc = np.asarray(scores, dtype=np.float64)
    out = np.zeros(q.shape, dtype=np.float64)
    for e in range(3):
        o = routed_attention(q, k, v, np.full(H, e, dtype=np.int32), model=model, latent=latent, tile=tile,
                             window=window, gi=gi, t_text=t_text, t_eff=t_eff)
        out += sc[:, :, e, None, None] * o
    return out


# ----------------------------------------------------------------------------------------------- A13
def ulysses_seq_to_head(shards: Sequence[np.ndarray]) -> List[np.ndarray]:
    """all_to_all_4D(x, scatter_idx=1, gather_idx=2) for every rank at once.

    shards[r] = (B,H,S/P,D) held by rank r.  Rank r ends with heads [r*H/P,(r+1)*H/P) and the
    rank-major concatenation of the sequence shards.  vorta/ulysses/utils.py:61-91."""
    P = len(shards)
    hl = shards[0].shape[1] // P
    return [np.concatenate([shards[src][:, r * hl:(r + 1) * hl] for src in range(P)], axis=2) for r in range(P)]


def ulysses_head_to_seq(shards: Sequence[np.ndarray]) -> List[np.ndarray]:
    """all_to_all_4D(x, scatter_idx=2, gather_idx=1): exact inverse.  vorta/ulysses/utils.py:33-59."""
    P = len(shards)
    sl = shards[0].shape[2] // P
    return [np.concatenate([shards[src][:, :, r * sl:(r + 1) * sl] for src in range(P)], axis=1) for r in range(P)]


def shrink_dim(x: np.ndarray, dim: int, rank: int, P: int) -> np.ndarray:
    """x.narrow(dim, rank*n/P, n/P).  vorta/ulysses/utils.py:218-223."""
    n = x.shape[dim] // P
    return np.take(x, np.arange(rank * n, (rank + 1) * n), axis=dim)


def all_gather_cat(parts: Sequence[np.ndarray], dim: int) -> np.ndarray:
    """rank-ordered concat.  vorta/ulysses/utils.py:135-146."""
    return np.concatenate(list(parts), axis=dim)


# ----------------------------------------------------------------------------------------------- N1 producer
def rms_norm(x: np.ndarray, weight: Optional[np.ndarray], eps: float, axis_size: Optional[int] = None) -> np.ndarray:
    """x * rsqrt(mean(x^2, -1) + eps) * weight  (hunyuan.py:69-72 attn.norm_q / norm_k, wan.py:86-89;
    [ext] the modules are diffusers / torch RMSNorm)."""
    x = np.asarray(x, dtype=np.float64)
    y = x / np.sqrt((x * x).mean(-1, keepdims=True) + eps)
    return y if weight is None else y * np.asarray(weight, dtype=np.float64)


def rope_interleaved(x: np.ndarray, cos: np.ndarray, sin: np.ndarray) -> np.ndarray:
    """Rotation of adjacent pairs: out = x*cos + rot(x)*sin, rot(x)[2i] = -x[2i+1], rot(x)[2i+1] = x[2i].
    hunyuan.py:97-98 ([ext] diffusers apply_rotary_emb(use_real=True, use_real_unbind_dim=-1)); identical to the
    complex product of wan.py:34-37 with cos/sin = Re/Im of the frequencies repeated per pair.  x: (...,S,D)."""
    x = np.asarray(x, dtype=np.float64)
    xr = x.reshape(x.shape[:-1] + (-1, 2))
    rot = np.stack([-xr[..., 1], xr[..., 0]], -1).reshape(x.shape)
    return x * cos + rot * sin


# ----------------------------------------------------------------------------------------------- misc
def pixel_to_token(n_pixel: int, ratio: int) -> int:
    """vorta/patch/utils.py:84-92."""
    n, mod = divmod(n_pixel, ratio)
    if mod == 0:
        return n
    if mod == 1:
        return n + 1
    raise ValueError(f"Number of pixel {n_pixel} is not a multiple of pixel2token {ratio}.")


def video_to_latent(video: Triple, temporal: int = 4, spatial: int = 16) -> Triple:
    """hunyuan_pixel2token == wan_pixel2token: /4 temporal, /(8*2) spatial.  vorta/patch/utils.py:59-95."""
    return (pixel_to_token(video[0], temporal), pixel_to_token(video[1], spatial), pixel_to_token(video[2], spatial))


# ----------------------------------------------------------------------------------------------- algorithmic work
def flops_full(S: int, t_eff: int, D: int = 128) -> float:
    return 4.0 * (S + t_eff) ** 2 * D


def flops_lowres(S_low: int, t_eff: int, D: int = 128) -> float:
    return 4.0 * (S_low + t_eff) ** 2 * D


def flops_sliding(S: int, tok: int, n_kv_tiles: int, t_eff: int, D: int = 128) -> float:
    return 4.0 * D * (S * (n_kv_tiles * tok + t_eff) + t_eff * (S + t_eff))


# ----------------------------------------------------------------------------------------------- fp8 (e4m3) path
# BASELINE.json configs[4] names an "fp8 MFMA QK^T/PV path"; the reference has no fp8 code, so there is nothing to
# restate from it.  What follows restates THIS build's own definition of that path (include/vorta_hip.h:
# vorta_fp8_quantize_qkv, vorta_attn_fwd_fp8) so the kernels can be checked rounding point for rounding point:
# the operands are e4m3 values, the scores are exact, the probabilities are rounded to e4m3 at the same reference
# points the kernel uses, everything else is float64.  With round_p=False the same code is plain softmax attention
# on the dequantised operands (and is checked against `_softmax_attend` in tests/test_oracle_fp8.py).
E4M3_MAX = 448.0


def e4m3_round(x: np.ndarray) -> np.ndarray:
    """Round to the nearest OCP e4m3fn value (ties to even), saturating at +-448.  Normal numbers have 3 mantissa
    bits (spacing 2^(e-3), e >= -6); below 2^-6 the spacing is 2^-9."""
    x = np.asarray(x, dtype=np.float64)
    a = np.minimum(np.abs(x), E4M3_MAX)
    with np.errstate(divide="ignore"):
        e = np.floor(np.log2(np.where(a > 0, a, 1.0)))
    e = np.maximum(e, -6.0)
    q = np.exp2(e - 3.0)
    r = np.rint(a / q) * q  # np.rint rounds half to even
    return np.copysign(np.minimum(r, E4M3_MAX), x)


def e4m3_encode(x: np.ndarray) -> np.ndarray:
    """float (already on the e4m3 grid or not) -> uint8 bit patterns (sign, 4-bit exponent bias 7, 3-bit mantissa)."""
    r = e4m3_round(x)
    a = np.abs(r)
    with np.errstate(divide="ignore"):
        e = np.floor(np.log2(np.where(a > 0, a, 1.0)))
    sub = a < 2.0 ** -6
    ebits = np.where(sub, 0, e + 7).astype(np.int64)
    mant = np.where(sub, np.rint(a * 512.0), np.rint((a / np.exp2(e) - 1.0) * 8.0)).astype(np.int64)
    return ((np.signbit(r).astype(np.int64) << 7) | (ebits << 3) | mant).astype(np.uint8)


def e4m3_decode(b: np.ndarray) -> np.ndarray:
    b = np.asarray(b).astype(np.int64)
    s, e, m = b >> 7, (b >> 3) & 15, b & 7
    v = np.where(e == 0, m * 2.0 ** -9, (1.0 + m / 8.0) * np.exp2(e - 7.0))
    v = np.where((e == 15) & (m == 7), np.nan, v)
    return np.where(s == 1, -v, v)


def fp8_center_rows(n_tokens: int, heads: int, seg_len: int = 0, tail_first: int = 0, tail_len: int = 0) -> list:
    """rows whose mean is the key centre of each head (fp8_quant.hip: fp8_kmean_kernel): TOKENS s = i * stride of the head
    with an odd stride ~ tokens / 1024, mapped to physical rows -- the identity for (H,S,D) views; in the segmented layout
    token s of head h is row ((s // seg_len) * heads + h) * seg_len + s % seg_len, the tail tokens follow at
    tail_first + h * seg_len."""
    if seg_len <= 0:
        stride = max(1, n_tokens // 1024) | 1
        return [np.arange(0, n_tokens, stride)] * heads
    has_tail = tail_first > 0 or tail_len > 0
    data_rows = tail_first if has_tail else n_tokens
    video = (data_rows // seg_len // heads) * seg_len
    per_head = video + (tail_len if has_tail else 0)
    stride = max(1, per_head // 1024) | 1
    s = np.arange(0, per_head, stride)
    rows = []
    for h in range(heads):
        r = np.where(s < video, ((s // seg_len) * heads + h) * seg_len + s % seg_len, tail_first + h * seg_len + (s - video))
        rows.append(r)
    return rows


def fp8_quantize_qkv(q: np.ndarray, k: np.ndarray, v: np.ndarray, scale: Optional[float] = None,
                     v_per_head: bool = False, k_center: Optional[np.ndarray] = None) -> dict:
    """include/vorta_hip.h vorta_fp8_quantize_qkv on (H,S,D) arrays holding bf16/fp16-representable values.
    The multipliers are computed in float32 like the kernel (fp8_quant.hip: fp8_scales_kernel), the products too.
    `k_center` (H,D) float32: the centre subtracted from the keys (flags bit1; any vector is legitimate -- softmax over
    the keys does not see it -- so tests pass the kernel's own, after checking it against `fp8_center_rows`).
    The abs-max of q and of the centred k is taken over the head's SAMPLE (the tokens of `fp8_center_rows`: the multipliers
    of q and k only balance the two operand ranges); v's per-channel abs-max is over every token."""
    f32 = np.float32
    q, k, v = (np.asarray(a, dtype=f32) for a in (q, k, v))
    if k_center is not None:
        k = (k - np.asarray(k_center, f32)[:, None, :]).astype(f32)
    D = q.shape[-1]
    c0 = f32(f32(1.0 / np.sqrt(D) if scale is None else scale) * f32(1.4426950408889634))
    rows = fp8_center_rows(q.shape[1], q.shape[0])
    mq = np.stack([np.abs(q[h, rows[h]]).max() for h in range(q.shape[0])]).astype(f32)
    mk = np.stack([np.abs(k[h, rows[h]]).max() for h in range(k.shape[0])]).astype(f32)
    t = np.ones_like(mq)
    ok = (mq > 0) & (mk > 0)
    t[ok] = np.sqrt((mk[ok] / (c0 * mq[ok]).astype(f32)).astype(f32)).astype(f32)
    qmul, kmul = (c0 * t).astype(f32), (f32(1.0) / t).astype(f32)
    mv = np.abs(v).max(1)  # (H,D)
    if v_per_head:
        mv = np.broadcast_to(mv.max(1, keepdims=True), mv.shape).copy()
    with np.errstate(divide="ignore"):
        vmul = np.where(mv > 0, f32(240.0) / mv, f32(0.0)).astype(f32)
    v_descale = (mv / f32(240.0)).astype(f32)
    clamp = lambda a: np.clip(a, -E4M3_MAX, E4M3_MAX)
    q8 = e4m3_encode(clamp((q * qmul[:, None, None]).astype(f32)))
    k8 = e4m3_encode(clamp((k * kmul[:, None, None]).astype(f32)))
    v8 = e4m3_encode(clamp((v * vmul[:, None, :]).astype(f32)))
    return dict(q8=q8, k8=k8, v8=v8, qmul=qmul, kmul=kmul, vmul=vmul, v_descale=v_descale)


# relative slack of the kernel's fp32 arithmetic against this float64 restatement: the MFMA accumulates 128 products
# in fp32 (flips were observed up to 5e-5 from a midpoint, tools/dbg/fp8_err.py) and v_exp_f32 is good to ~1 ulp
_FP8_AMBIG = 1e-4


def _fp8_flash_rows(Q: np.ndarray, K: np.ndarray, V: np.ndarray, blk_lo: int, blk_hi: int, p_bias: float, defer: float,
                    round_p: bool, block: int = 64, ambiguous: Optional[np.ndarray] = None, p_mode: str = "rne"):
    """One wave of the fp8 kernel (<= 32 query rows, all sharing the reference-point decisions) over key blocks
    [blk_lo, blk_hi).  Q (n,D), K/V (n_kv,D) are decoded e4m3 values; Q . K is the score in the exp2 domain.
    Returns unnormalised O (n,D), row sums l (n,) and reference points m (n,), all relative to 2^p_bias.
    attn_fwd_fp8.hip: the first block fixes m at its row max; later a block whose offset row max exceeds `defer` for
    ANY row of the wave moves every row's reference to max(its own block max, its reference).
    `ambiguous` (n,) float, optional: accumulates, per row, the P' of every probability that lies within fp32 noise of an
    e4m3 rounding midpoint (a correct kernel may round those the other way); set to +inf for the whole wave when a
    block max is within noise of `defer` (the reference point itself may differ)."""
    n_kv = K.shape[0]
    z_all = Q @ K[blk_lo * block:min(blk_hi * block, n_kv)].T
    n = Q.shape[0]
    O = np.zeros((n, V.shape[1]))
    l = np.zeros(n)
    m = None
    if p_mode == "rne_mx":  # the mixed kernel (attn_fwd_mx.hip) since ABI 7: exp2 + round-to-nearest e4m3 under the same block scales
        return _i8_mx_flash_rows(z_all, V, blk_lo, blk_hi, p_bias, defer, round_p, block, ambiguous, rne=True)
    if p_mode == "mx":
        # the kernel's byte-domain offset is ~1.5 2^23 x 8 u (u = one integer score unit in the exp2 domain, the operands' last
        # column) rounded to fp32 twice (the row's offset, then the tile's): a probability's y may sit that far from this
        # restatement's -- the width of the midpoint band of `ambiguous`
        slack = max(2e-3, 2.0 ** -22 * 12582912.0 * 8.0 * abs(float(Q[0, -1])))
        return _i8_mx_flash_rows(z_all, V, blk_lo, blk_hi, p_bias, defer, round_p, block, ambiguous, slack)
    for j in range(blk_lo, blk_hi):
        lo, hi = (j - blk_lo) * block, min((j - blk_lo + 1) * block, z_all.shape[1])
        z = z_all[:, lo:hi]
        if m is None:
            m = z.max(1)
        else:
            mx = (z - m[:, None]).max(1)
            if ambiguous is not None and (np.abs(mx - defer) < 1e-4).any():
                ambiguous[:] = np.inf
            if (mx > defer).any():
                g = np.maximum(mx, 0.0)
                a = np.exp2(-g)
                O *= a[:, None]
                l *= a
                if ambiguous is not None:
                    ambiguous *= a
                m = m + g
        P = np.exp2(z - m[:, None] + p_bias)
        if round_p and p_mode == "direct":
            # the int8-score kernel up to ABI 6: the e4m3 BYTE is rint(8 x + 56), x = log2 P' (one v_cvt_pk_u8_f32, saturating at
            # 0): exponent field = integer part of x, mantissa = linear interpolation of its fraction
            y = 8.0 * (z - m[:, None] + p_bias) + 56.0
            if ambiguous is not None:
                near = np.rint(y + 2e-3) != np.rint(y - 2e-3)
                ambiguous += (P * near).sum(1)
            with np.errstate(invalid="ignore"):
                P = e4m3_decode(np.clip(np.rint(np.where(np.isfinite(y), y, 0.0)), 0, 126).astype(np.int64))
        elif round_p:
            if ambiguous is not None:
                near = e4m3_round(P * (1 + _FP8_AMBIG)) != e4m3_round(P * (1 - _FP8_AMBIG))
                ambiguous += (P * near).sum(1)
            P = e4m3_round(P)
        l += P.sum(1)
        O += P @ V[j * block:j * block + (hi - lo)]
    return O, l, m


I8_YMID = 120.0   # a tile's largest byte-domain value lands in [116, 124] (0x7E = 126 = 448 is the largest e4m3)
I8_EMIN = -100.0  # block exponent >= -100 (no upper clamp: the reference point moves under a tile that lies far above it)
# a scale block of the P V MFMA's B operand = one query row x the 32 consecutive keys of one key tile of the 64-key block
# (8-bit operands: bytes 16 s ... 16 s + 15 of both lanes of a column; tools/probe_mx_scale.hip)
_I8_GROUP = np.arange(64) >> 5


def _i8_mx_flash_rows(z_all: np.ndarray, V: np.ndarray, blk_lo: int, blk_hi: int, p_bias: float, etrig: float, round_p: bool,
                      block: int, ambiguous: Optional[np.ndarray], slack: float = 2e-3, rne: bool = False):
    """One wave of vorta_attn_fwd_i8 (ABI 7, attn_fwd_i8.hip): MX-SCALED probabilities.  Byte domain y = 8 (z - m + p_bias) + 56
    against the row's reference point m (the maximum of its first block).  Per 64-key block and per KEY TILE of 32 consecutive
    keys (`_I8_GROUP`; keys past the end of the list repeat the last one: the kernel clamps its rows), for every query row: the
    block exponent e = max(rint((max y - 120) / 8), -100), the bytes rint(y - 8 e) (saturating at 0; masked keys 0), the
    probability = e4m3(byte) 2^e: the scale goes to the P V MFMA as the B operand's block scale.  The reference point moves
    only when some (row, tile) of the wave has e > etrig: then every row moves by max(its larger e, 0) whole binades (bytes
    unchanged).  `ambiguous`: as `_fp8_flash_rows`, plus -- where a tile's exponent is within noise of its neighbour -- the
    mass of its probabilities below byte 16 (one exponent further they decode through e4m3's linear subnormals, not to the
    same values).
    `rne=True`: the MIXED kernel's form of the same scheme (attn_fwd_mx.hip): scores c = z - m + p_bias in the exp2 domain, e =
    rint(max(max c + 64, 0) - 72) per (row, tile) -- the kernel takes the tile maximum over scores shifted up by 64 binades and
    reads a tile that is negative throughout as 0 -- probability = e4m3_rne(2^(c - e)) 2^e (v_exp_f32, then
    v_cvt_scalef32_pk_fp8_f32: round to nearest even with subnormals; the tile's largest lands in [2^7.5, 2^8.5))."""
    assert block == 64
    n = z_all.shape[0]
    O = np.zeros((n, V.shape[1]))
    l = np.zeros(n)
    m = None
    for j in range(blk_lo, blk_hi):
        lo, hi = (j - blk_lo) * block, min((j - blk_lo + 1) * block, z_all.shape[1])
        z = z_all[:, lo:hi]
        if m is None:
            m = z.max(1)
        y = 8.0 * (z - m[:, None] + p_bias) + 56.0
        ypad = np.concatenate([y, np.repeat(y[:, -1:], block - (hi - lo), 1)], 1) if hi - lo < block else y
        ymx = np.stack([ypad[:, _I8_GROUP == g].max(1) for g in (0, 1)], 1)  # (n, 2)
        if rne:
            cmx = (ymx - 56.0) / 8.0  # the tile maximum in the exp2 domain
            t = np.maximum(np.where(np.isfinite(cmx), cmx, -1e30) + 64.0, 0.0) - 72.0
            e = np.rint(t)
            e_near = np.abs(np.abs(t - np.floor(t)) - 0.5) < 1e-4
        else:
            t = (ymx - I8_YMID) / 8.0
            with np.errstate(invalid="ignore"):
                e = np.maximum(np.rint(np.where(np.isfinite(t), t, I8_EMIN)), I8_EMIN)
            e_near = np.abs(np.abs(t - np.floor(t)) - 0.5) < slack / 4.0  # a correct kernel may round the other way
        if j > blk_lo:
            # (a trigger within noise of its threshold needs no mark: moving the reference point changes power-of-two scales on
            # both sides of the accumulation, not one byte and not one value)
            if (e > etrig).any():
                g = np.maximum(e.max(1), 0.0)
                a = np.exp2(-g)
                O *= a[:, None]
                l *= a
                if ambiguous is not None:
                    ambiguous *= a
                m = m + g
                e = np.maximum(e - g[:, None], I8_EMIN)
                y = y - 8.0 * g[:, None]
        ek = e[:, _I8_GROUP[:hi - lo]]  # (n, keys): each key's lane exponent
        yp = y - 8.0 * ek
        Pex = np.exp2(z - m[:, None] + p_bias)
        if round_p and rne:
            P8x = np.exp2(np.where(np.isfinite(yp), (yp - 56.0) / 8.0, -np.inf))
            assert P8x.max() <= 448.0, P8x.max()  # (the hardware conversion would write NaN)
            P = e4m3_round(P8x) * np.exp2(ek)
            if ambiguous is not None:
                near = e4m3_round(P8x * (1 + _FP8_AMBIG)) != e4m3_round(P8x * (1 - _FP8_AMBIG))
                ambiguous += (Pex * near).sum(1)
                ambiguous += (Pex * ((P8x < 2.0 ** -5) & e_near[:, _I8_GROUP[:hi - lo]])).sum(1)
        elif round_p:
            with np.errstate(invalid="ignore"):
                byte = np.clip(np.rint(np.where(np.isfinite(yp), yp, 0.0)), 0, 126).astype(np.int64)
            P = e4m3_decode(byte) * np.exp2(ek)
            if ambiguous is not None:
                near = np.rint(yp + slack) != np.rint(yp - slack)
                ambiguous += (Pex * near).sum(1)
                ambiguous += (Pex * ((byte < 16) & e_near[:, _I8_GROUP[:hi - lo]])).sum(1)
        else:
            P = Pex
        l += P.sum(1)
        O += P @ V[j * block:j * block + (hi - lo)]
    return O, l, m


def fp8_attn_launch(q: np.ndarray, k: np.ndarray, v: np.ndarray, out: np.ndarray, v_descale: np.ndarray, *,
                    n_q: int, n_kv: int, q_rows: Optional[np.ndarray] = None, q_row_offset: int = 0,
                    q_group_len: int = 0, q_valid: Optional[int] = None, kv_rows: Optional[np.ndarray] = None,
                    kv_row_offset: int = 0, dup_rows: Optional[np.ndarray] = None, n_dup_pos: int = 0, n_splits: int = 1,
                    p_bias: float = 5.0, defer: float = 3.0, round_p: bool = True,
                    ambiguous: Optional[np.ndarray] = None,
                    q_group_bounds: Optional[Sequence[Tuple[int, int]]] = None,
                    wave_filter=None, wave_operands=None, p_mode: str = "rne") -> None:
    """include/vorta_hip.h vorta_attn_fwd_fp8 for ONE head: q,k,v (rows,D) decoded e4m3 values, `out` (rows,D) is
    written in place (rows named by q_rows / dup_rows only).  kv_rows: (n_kv,) or (n_groups, n_kv).
    `ambiguous` (rows,) float, optional: per output row, the total normalised probability of the keys whose e4m3
    rounding is within fp32 noise of a midpoint (see `_fp8_flash_rows`; 0 for most rows, inf where a reference point
    is in doubt).  Each of those may move by one e4m3 step (<= 2^-3 relative), so a correct kernel differs from this
    restatement by at most 2^-3 * that * (|v| + |o|) on top of its accumulation error.
    `wave_filter(group, first_position_in_group) -> bool`, optional: restate only the waves it accepts (sampled checks at
    sizes where every wave would take hours); rows of the other waves are left untouched.
    `wave_operands(query_rows, key_rows) -> (Qw, Kw)`, optional: the score operands of ONE wave (vorta_attn_fwd_i8: the query
    scale, hence the rounded key biases, belong to the wave -- `i8_wave_operands`); q and k are then not read.
    `p_mode="direct"`: the probabilities' bytes are rint(8 log2 P' + 56) (see `_fp8_flash_rows`); `p_mode="mx"`: vorta_attn_fwd_i8
    since ABI 7 -- one power-of-two scale per lane and block (`_i8_mx_flash_rows`; `defer` is then the trigger in binades,
    24 in the kernel's default)."""
    q_valid = n_q if q_valid is None else q_valid
    glen = q_group_len if q_group_len > 0 else n_q
    if q_group_bounds is None:  # equal groups; else [start, end) of every group (vorta_attn_args.q_block_table)
        q_group_bounds = [(g * glen, min((g + 1) * glen, n_q)) for g in range(-(-n_q // glen))]
    nblk = -(-n_kv // 64)
    bps = -(-nblk // n_splits)
    for g, (g0, g1) in enumerate(q_group_bounds):
        pos = np.arange(g0, g1)
        rows = q_rows[pos] if q_rows is not None else q_row_offset + pos
        if kv_rows is None:
            kr = kv_row_offset + np.arange(n_kv)
        else:
            kr = (kv_rows[g] if kv_rows.ndim == 2 else kv_rows)[:n_kv]
        if wave_filter is not None and not any(wave_filter(g, w0) for w0 in range(0, len(pos), 32)):
            continue
        Kg, Vg = (k[kr] if wave_operands is None else None), v[kr]
        for w0 in range(0, len(pos), 32):  # one wave = 32 consecutive positions of the group
            if wave_filter is not None and not wave_filter(g, w0):
                continue
            sl = slice(w0, min(w0 + 32, len(pos)))
            if wave_operands is None:
                Qw = q[rows[sl]]
            else:
                Qw, Kg = wave_operands(rows[sl], kr)
            parts, ambs = [], []
            for s in range(n_splits):
                if s * bps < nblk:
                    ambs.append(np.zeros(Qw.shape[0]) if ambiguous is not None else None)
                    parts.append(_fp8_flash_rows(Qw, Kg, Vg, s * bps, min((s + 1) * bps, nblk), p_bias, defer, round_p,
                                                 ambiguous=ambs[-1], p_mode=p_mode))
            mm = np.max([p[2] for p in parts], axis=0)
            O = sum(p[0] * np.exp2(p[2] - mm)[:, None] for p in parts)
            l = sum(p[1] * np.exp2(p[2] - mm) for p in parts)
            res = np.where((pos[sl] < q_valid)[:, None] & (l > 0)[:, None], O / np.where(l > 0, l, 1.0)[:, None], 0.0)
            res = res * v_descale[None, :]
            out[rows[sl]] = res
            if ambiguous is not None:
                amb = sum(a * np.exp2(p[2] - mm) for a, p in zip(ambs, parts)) / np.where(l > 0, l, 1.0)
                ambiguous[rows[sl]] = amb
            if dup_rows is not None:
                for i, p_ in enumerate(pos[sl]):
                    if p_ < n_dup_pos:
                        out[dup_rows[p_]] = res[i]
                        if ambiguous is not None:
                            ambiguous[dup_rows[p_]] = amb[i]


# ---------------------------------------------------------------------------------------------- int8 scores (ABI 6-7)
# No reference counterpart either: this restates include/vorta_hip.h vorta_i8_quantize_k (csrc/i8_quant.hip) and the query
# conversion at the head of vorta_attn_fwd_i8 (csrc/attn_fwd_i8.hip) operation for operation in float32, so the kernels can be
# held to it bit for bit; the attention itself is `fp8_attn_launch` with `wave_operands` (the scores of a wave are
# u (q8 . k8 + seed), an exact integer dot product) and `p_mode="mx"` (the probabilities' e4m3 bytes are rint(8 x + 56 - 8 e)
# with one block exponent e per lane: `_i8_mx_flash_rows`).
I8_MAGIC_LIMIT = 2000000.0  # |bias| in integer score units is clamped to this (the int32 accumulator is read as a float)


def i8_sample_tokens(n_tokens: int) -> np.ndarray:
    """tokens of a head's sample: i * stride, stride = (n_tokens // 1024) | 1 (odd, >= 1)"""
    stride = max(1, n_tokens // 1024) | 1
    return np.arange(0, n_tokens, stride)


def i8_quantize_k(q: np.ndarray, k: np.ndarray, smooth: bool = True, center: bool = True) -> dict:
    """(H,S,D) arrays of bf16 / fp16-representable values -> dict(k8 int8 (H,S,D), k_bias f32 (H,S), q_prep f32 (H,2,D) = cq | s,
    k_head_scale f32 (H,), center_k f32 (H,D)).  Sample statistics with the kernel's summation order: row lane rl of 64 adds
    its samples rl, rl + 64, ... in order, the 64 partial sums are added pairwise at distance 32, 16, ..., 1; the bias dot
    product: 8 channels per lane in order, the 16 lanes pairwise at distance 1, 2, 4, 8; every product and sum rounded to
    float32 on its own."""
    f32 = np.float32
    q, k = np.asarray(q, dtype=f32), np.asarray(k, dtype=f32)
    H, S, D = k.shape
    tok = i8_sample_tokens(S)
    n = f32(len(tok))
    ck, cq, smo = np.zeros((H, D), f32), np.zeros((H, D), f32), np.ones((H, D), f32)
    for h in range(H):
        ks, qs = k[h, tok], q[h, tok]
        part = np.zeros((4, 64, D), f32)
        for i in range(len(tok)):
            rl = i % 64
            part[0, rl] = part[0, rl] + ks[i]
            part[1, rl] = part[1, rl] + ks[i] * ks[i]
            part[2, rl] = part[2, rl] + qs[i]
            part[3, rl] = part[3, rl] + qs[i] * qs[i]
        off = 32
        while off > 0:
            part[:, :off] = part[:, :off] + part[:, off:2 * off]
            off >>= 1
        mean_k, mean_q = part[0, 0] / n, part[2, 0] / n
        var_k = part[1, 0] / n - mean_k * mean_k
        var_q = part[3, 0] / n - mean_q * mean_q
        sv = np.ones(D, f32)
        ok = (var_k > 0) & (var_q > 0)
        if smooth:
            with np.errstate(divide="ignore", invalid="ignore"):
                s_all = np.sqrt(np.sqrt((var_k / var_q).astype(f32)).astype(f32)).astype(f32)
            sv[ok] = np.minimum(np.maximum(s_all[ok], f32(0.125)), f32(8.0))
        if center:
            ck[h], cq[h] = mean_k, mean_q
        smo[h] = sv
    inv_s = (f32(1.0) / smo).astype(f32)
    d = (k - ck[:, None, :]).astype(f32)
    kt = (d * inv_s[:, None, :]).astype(f32)
    am = np.abs(kt).reshape(H, -1).max(-1)
    with np.errstate(divide="ignore"):
        inv = np.where(am > 0, f32(127.0) / am, f32(0.0)).astype(f32)
    k8 = np.clip(np.rint((kt * inv[:, None, None]).astype(f32)), -127, 127).astype(np.int8)
    pr = (cq[:, None, :] * d).astype(f32).reshape(H, S, 16, 8)
    lane = np.zeros((H, S, 16), f32)
    for e in range(8):
        lane = lane + pr[..., e]
    while lane.shape[-1] > 1:
        lane = lane[..., 0::2] + lane[..., 1::2]
    k_bias = (lane[..., 0] * inv[:, None]).astype(f32)
    k_head_scale = np.where(am > 0, (am * f32(1.0 / 127.0)).astype(f32), f32(1.0)).astype(f32)
    return dict(k8=k8, k_bias=k_bias, q_prep=np.stack([cq, smo], 1), k_head_scale=k_head_scale, center_k=ck)


def i8_wave_operands(q_rows: np.ndarray, q_prep: np.ndarray, sk: float, k8: np.ndarray, k_bias: np.ndarray,
                     scale: Optional[float] = None):
    """What ONE wave of vorta_attn_fwd_i8 multiplies: q_rows (n <= 32, D) 16-bit-representable query rows of the wave, q_prep
    (2, D) = cq | s of the head, sk its key scale, k8 (n_kv, D) / k_bias (n_kv,) the keys of the wave's list.  The wave's
    query scale is the abs-max over ALL its rows / 127; the accumulator starts from rint(k_bias / sq) (clamped).  Returns
    (Q (n, D+1), K (n_kv, D+1)) float64 with Q @ K.T = u (q8 . k8 + seed): the exp2-domain scores."""
    f32 = np.float32
    q = np.asarray(q_rows, dtype=f32)
    D = q.shape[-1]
    c0 = f32(f32(1.0 / np.sqrt(D) if scale is None else scale) * f32(1.4426950408889634))
    qt = ((q - np.asarray(q_prep[0], f32)[None, :]).astype(f32) * np.asarray(q_prep[1], f32)[None, :]).astype(f32)
    am = f32(np.abs(qt).max())
    flat = not (am >= f32(2.0 ** -12))  # a wave on the head's centre: q8 = 0, unit scale (the bias term alone makes its scores)
    inv = f32(1.0) if flat else f32(127.0) / am
    q8 = np.clip(np.rint((qt * (f32(0.0) if flat else inv)).astype(f32)), -127, 127).astype(np.float64)
    sq = f32(1.0) if flat else (am * f32(1.0 / 127.0))
    u = f32(f32(sq * c0) * f32(sk))
    # (ABI 7: ONE rounding of the exact product -- the kernel's fused multiply-add into the binade of 1.5 2^23 -- then the clamp)
    seed = np.clip(np.rint(np.asarray(k_bias, np.float64) * np.float64(inv)), -I8_MAGIC_LIMIT, I8_MAGIC_LIMIT)
    Q = np.concatenate([q8 * float(u), np.full((q8.shape[0], 1), float(u))], 1)
    K = np.concatenate([np.asarray(k8, np.float64), seed[:, None]], 1)
    return Q, K
