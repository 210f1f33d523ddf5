"""GPU: the attention-processor classes (the drop-in boundary) end to end -- projections in torch, everything
between post-RoPE q,k,v and the output projection in libvorta_hip.so -- against the reference's golden output
(Wan, whole processor call) and against a float64 restatement built on the oracle (Hunyuan)."""
import math
import os
import socket

import numpy as np
import pytest
import torch
from torch import nn

from oracle import vorta_oracle as O
from _util import dev, rel_fro

pytestmark = pytest.mark.gpu

LATENT, TILE, WINDOW, GROUP = (8, 6, 8), (2, 3, 4), (3, 3, 3), (2, 3, 2)
S = 8 * 6 * 8
H = 6


# --------------------------------------------------------------------------------------------- Wan, golden G8
class _RmsAcrossHeadsPadded(nn.Module):
    """RMSNorm over the TRUE channels of a zero-padded (.., H*128) activation (padding holds zeros)."""

    def __init__(self, weight_padded, n_true, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(weight_padded, requires_grad=False)
        self.n_true, self.eps = n_true, eps

    def forward(self, x):
        ms = x.float().pow(2).sum(-1, keepdim=True) / self.n_true
        return (x.float() * torch.rsqrt(ms + self.eps) * self.weight.float()).to(x.dtype)


def _pad_rows(w, d_true, d_pad):  # (H*d_true, in) -> (H*d_pad, in), each head's rows followed by zeros
    out = np.zeros((H * d_pad,) + w.shape[1:], dtype=w.dtype)
    for h in range(H):
        out[h * d_pad:h * d_pad + d_true] = w[h * d_true:(h + 1) * d_true]
    return out


class _WanFakeAttn(nn.Module):
    """The golden fixture's fake attention module (H=6, head dim 16) embedded in head dim 128 by zero padding:
    the same function, sized for the kernels.  q is pre-scaled by sqrt(128/16) so that the kernels' 1/sqrt(128)
    equals the reference's 1/sqrt(16)."""

    def __init__(self, g, dtype):
        super().__init__()
        self.heads = H
        self.add_k_proj = None
        d, D = 16, 128

        def lin(name, scale=1.0):
            w, b = g[f"wan_w_{name}_weight"], g[f"wan_w_{name}_bias"]
            m = nn.Linear(H * d, H * D)
            m.weight.data = torch.tensor(_pad_rows(w, d, D))
            m.bias.data = torch.tensor(_pad_rows(b, d, D))
            return m

        self.to_q, self.to_k, self.to_v = lin("to_q"), lin("to_k"), lin("to_v")
        self.norm_q = _RmsAcrossHeadsPadded(torch.tensor(_pad_rows(g["wan_w_norm_q_weight"], d, D)) * math.sqrt(D / d), H * d)
        self.norm_k = _RmsAcrossHeadsPadded(torch.tensor(_pad_rows(g["wan_w_norm_k_weight"], d, D)), H * d)
        out = nn.Linear(H * D, H * d)
        out.weight.data = torch.tensor(_pad_rows(g["wan_w_to_out_0_weight"].T.copy(), d, D).T.copy())
        out.bias.data = torch.tensor(g["wan_w_to_out_0_bias"])
        self.to_out = nn.ModuleList([out, nn.Identity()])
        self.to(dev()).to(dtype)


def _wan_kwargs():
    from vorta_amd.patch import prepare_wan_self_attn_kwargs
    return prepare_wan_self_attn_kwargs(dict(latent_shape=LATENT, window_size=WINDOW, tile_size=TILE,
                                             lowres_window_size=GROUP, lowres_reduction_rate=0.5), dev())


@pytest.mark.parametrize("tau", [0.3, 0.9])
def test_wan_triple_eval_whole_call_golden(golden, tau):
    from vorta_amd.attention import WanAttnProcessorTripleEval
    g = golden("g8_eval_calls")
    dtype = torch.bfloat16
    attn = _WanFakeAttn(g, dtype)
    hidden = torch.tensor(g["wan_hidden"]).to(dtype).to(dev())
    proc = WanAttnProcessorTripleEval(check_input=True)
    y = proc(attn, hidden, None, None, None, tau_sparse=tau, routing_score=torch.tensor(g["routing_score"]).to(dev()),
             **_wan_kwargs())
    gold = g[f"wan_out_tau{int(tau * 10)}"]
    assert y.shape == gold.shape
    err = np.abs(y.float().cpu().numpy() - gold)
    # bf16 weights/activations through three linears + norm: stated tolerance 3e-2 abs, 2e-2 relative Frobenius
    assert rel_fro(y.float().cpu().numpy(), gold) < 2e-2
    assert (err.max(-1) > 3e-2).mean() < 0.02


def test_wan_triple_eval_batch_follows_item_zero_routes(golden):
    """B > 1: the reference routes every batch item by item 0's scores (wan.py:388-416); here a batch is a loop over its
    items with those routes -- equal to the items run one by one with item 0's score tensor"""
    from vorta_amd.attention import WanAttnProcessorTripleEval
    g = golden("g8_eval_calls")
    dtype = torch.bfloat16
    attn = _WanFakeAttn(g, dtype)
    hidden = torch.tensor(g["wan_hidden"]).to(dtype).to(dev())
    torch.manual_seed(3)
    batch = torch.cat([hidden, (hidden.float() + 0.3 * torch.randn_like(hidden.float())).to(dtype)], dim=0)
    score0 = torch.tensor(g["routing_score"]).to(dev())
    score = torch.cat([score0, torch.softmax(torch.randn_like(score0), -1)], dim=0)  # item 1's own scores are ignored
    proc = WanAttnProcessorTripleEval(check_input=True)
    y = proc(attn, batch, None, None, None, tau_sparse=0.3, routing_score=score, **_wan_kwargs())
    assert y.shape[0] == 2
    for b in range(2):
        yb = proc(attn, batch[b:b + 1], None, None, None, tau_sparse=0.3, routing_score=score0, **_wan_kwargs())
        # (the one-item call takes the fused qk-norm kernel, the batch the norm modules: bf16 rounding apart)
        assert rel_fro(y[b:b + 1].float().cpu().numpy(), yb.float().cpu().numpy()) < 2e-2


def test_wan_check_input_and_cross_attention(golden):
    from vorta_amd.attention import WanAttnProcessor2_0, WanAttnProcessorTripleEval
    g = golden("g8_eval_calls")
    dtype = torch.bfloat16
    attn = _WanFakeAttn(g, dtype)
    proc = WanAttnProcessorTripleEval(check_input=True)
    hidden = torch.tensor(g["wan_hidden"]).to(dtype).to(dev())
    with pytest.raises(ValueError):  # sequence does not match the latent grid (wan.py:181-184)
        proc(attn, hidden[:, :-8], None, None, None, tau_sparse=0.3,
             routing_score=torch.tensor(g["routing_score"]).to(dev()), **_wan_kwargs())
    # cross attention (Sq != Skv) goes through the dense kernel, and equals the dense processor
    enc = torch.randn((1, 40, 96), device=dev()).to(dtype)
    y1 = proc(attn, hidden, enc, None, None, tau_sparse=0.3, routing_score=None, **_wan_kwargs())
    y2 = WanAttnProcessor2_0()(attn, hidden, enc, None, None)
    assert torch.equal(y1, y2)
    with torch.no_grad():
        q, k, v, _ = proc._input_proj(attn, hidden, enc, None)
        ref = O.dense_attention(q.double().cpu().numpy(), k.double().cpu().numpy(), v.double().cpu().numpy())
        ref = torch.tensor(ref).permute(0, 2, 1, 3).flatten(2, 3).to(dtype).to(dev())
        want = attn.to_out[0](ref)
    y2 = y2.detach()
    assert rel_fro(y2.float().cpu().numpy(), want.float().cpu().numpy()) < 1e-2


# --------------------------------------------------------------------------------------------- Hunyuan vs oracle
class _HyFakeAttn(nn.Module):
    def __init__(self, hidden, dual, dtype, seed):
        super().__init__()
        torch.manual_seed(seed)
        D = 128
        self.heads = H
        self.to_q, self.to_k, self.to_v = (nn.Linear(hidden, H * D) for _ in range(3))
        self.norm_q, self.norm_k = nn.RMSNorm(D, eps=1e-6), nn.RMSNorm(D, eps=1e-6)
        if dual:
            self.add_q_proj, self.add_k_proj, self.add_v_proj = (nn.Linear(hidden, H * D) for _ in range(3))
            self.norm_added_q, self.norm_added_k = nn.RMSNorm(D, eps=1e-6), nn.RMSNorm(D, eps=1e-6)
            self.to_out = nn.ModuleList([nn.Linear(H * D, hidden), nn.Identity()])
            self.to_add_out = nn.Linear(H * D, hidden)
        else:
            self.add_q_proj = self.add_k_proj = self.add_v_proj = None
            self.norm_added_q = self.norm_added_k = None
            self.to_out = None
            self.to_add_out = None
        for m in self.modules():
            if isinstance(m, nn.RMSNorm):
                nn.init.uniform_(m.weight, 0.5, 1.5)
        self.to(dev()).to(dtype)


def _f64(t):
    return t.detach().double().cpu().numpy()


def _hy_reference(attn, hidden, enc, rope, experts, T, te, dual, qkv=None):
    """float64 restatement of hunyuan.py:544-608 with the oracle as the attention core.  With `qkv` given the
    attention runs on exactly those (already rounded) q,k,v, so coreset rankings cannot differ by rounding; the
    projections are then checked separately (returned as the third value)."""
    def lin(m, x):
        return x @ _f64(m.weight).T + _f64(m.bias)

    def rms(m, x):
        return x / np.sqrt((x * x).mean(-1, keepdims=True) + 1e-6) * _f64(m.weight)

    def heads(x):
        return x.reshape(1, x.shape[1], H, 128).transpose(0, 2, 1, 3)

    def rot(x):  # interleaved pairs (diffusers apply_rotary_emb, use_real_unbind_dim=-1)
        cos, sin = (_f64(r)[None, None] for r in rope)
        xr = x.reshape(*x.shape[:-1], -1, 2)
        xrot = np.stack([-xr[..., 1], xr[..., 0]], -1).reshape(x.shape)
        return x * cos + xrot * sin

    x = _f64(hidden)
    e = _f64(enc)
    if not dual:
        x = np.concatenate([x, e], 1)
    q, k, v = heads(lin(attn.to_q, x)), heads(lin(attn.to_k, x)), heads(lin(attn.to_v, x))
    q, k = rms(attn.norm_q, q), rms(attn.norm_k, k)
    if dual:
        q, k = rot(q), rot(k)
        eq, ek, ev = heads(lin(attn.add_q_proj, e)), heads(lin(attn.add_k_proj, e)), heads(lin(attn.add_v_proj, e))
        eq, ek = rms(attn.norm_added_q, eq), rms(attn.norm_added_k, ek)
        q, k, v = (np.concatenate(p, 2) for p in ((q, eq), (k, ek), (v, ev)))
    else:
        q = np.concatenate([rot(q[:, :, :S]), q[:, :, S:]], 2)
        k = np.concatenate([rot(k[:, :, :S]), k[:, :, S:]], 2)
    gi = O.group_info(LATENT, GROUP, 0.5)
    projected = (q, k, v)
    if qkv is not None:
        q, k, v = (_f64(x) for x in qkv)
    o = O.routed_attention(q, k, v, np.asarray(experts), model="hunyuan", latent=LATENT, tile=TILE, window=WINDOW,
                           gi=gi, t_text=T, t_eff=te)
    o = o.transpose(0, 2, 1, 3).reshape(1, S + T, H * 128)
    hid, en = o[:, :S], o[:, S:]
    if dual:
        hid, en = lin(attn.to_out[0], hid), lin(attn.to_add_out, en)
    return hid, en, projected


@pytest.mark.parametrize("dual", [True, False])
def test_hunyuan_triple_eval_vs_oracle(dual):
    from vorta_amd.attention import (HunyuanVideoFlashAttnProcessor, HunyuanVideoFlashAttnProcessorTripleEval,
                                     create_sliding_tile_attn_mask_func, get_group_info)
    dtype = torch.bfloat16
    hidden_dim, T, te = 64, 16, 11
    attn = _HyFakeAttn(hidden_dim, dual, dtype, seed=3 + dual)
    torch.manual_seed(5)
    hidden = torch.randn((1, S, hidden_dim), device=dev()).to(dtype)
    enc = torch.randn((1, T, hidden_dim), device=dev()).to(dtype)
    ang = torch.rand((S, 64), device=dev()) * 6.28
    rope = (ang.cos().repeat_interleave(2, dim=1), ang.sin().repeat_interleave(2, dim=1))
    mask = torch.zeros((1, 1, 1, S + T), dtype=torch.bool, device=dev())
    mask[..., :S + te] = True
    experts = [0, 1, 2, 2, 1, 0]
    score = torch.full((1, H, 3), 0.1, device=dev())
    for h, e in enumerate(experts):
        score[0, h, e] = 0.8
    kw = dict(lowres_group_info=get_group_info(LATENT, GROUP, 0.5, dev()), window_size=WINDOW, tile_size=TILE,
              latent_shape=LATENT,
              flex_attn_mask_func=create_sliding_tile_attn_mask_func(LATENT, WINDOW, TILE, T, te, dev()))
    proc = HunyuanVideoFlashAttnProcessorTripleEval(check_input=True)
    hid, en = proc(attn, hidden, enc, mask, rope, routing_score=score, tau_sparse=0.3, **kw)
    with torch.no_grad():
        qkv = proc._project(attn, hidden, enc, rope)[:3]
    ref_h, ref_e, projected = _hy_reference(attn, hidden, enc, rope, experts, T, te, dual, qkv=qkv)
    for got, want in zip(qkv, projected):  # steps 1-4: projections, qk-norm, RoPE, text concat (bf16 vs float64)
        assert rel_fro(got.float().cpu().numpy(), want) < 1e-2
    assert hid.shape == ref_h.shape and en.shape == ref_e.shape
    assert rel_fro(hid.float().cpu().numpy(), ref_h) < 1.5e-2 and rel_fro(en.float().cpu().numpy(), ref_e) < 1.5e-2
    if not dual:  # no output projection: padded text rows are exactly zero (hunyuan.py:176)
        assert torch.all(en[:, te:] == 0)
    # without the descriptor the text length is read from the mask (like hunyuan.py:169) and nothing changes
    kw2 = dict(kw, flex_attn_mask_func=None)
    hid2, en2 = proc(attn, hidden, enc, mask, rope, routing_score=score, tau_sparse=0.3, **kw2)
    assert torch.equal(hid, hid2) and torch.equal(en, en2)
    # tau above every score -> every head dense == the native-attention processor (its L stays on the device)
    hid3, en3 = proc(attn, hidden, enc, mask, rope, routing_score=score, tau_sparse=0.95, **kw)
    hid4, en4 = HunyuanVideoFlashAttnProcessor()(attn, hidden, enc, mask, rope)
    assert torch.equal(hid3, hid4) and torch.equal(en3, en4)
    ref_h, ref_e, _ = _hy_reference(attn, hidden, enc, rope, [0] * H, T, te, dual, qkv=qkv)
    assert rel_fro(hid4.float().cpu().numpy(), ref_h) < 1.5e-2


@pytest.mark.parametrize("dual", [True, False])
@pytest.mark.parametrize("precision", ["native", "fp8", "i8pv", "auto8"])
def test_hunyuan_processor_call_compiles_fullgraph(dual, precision):
    """the production processor call goes through torch.ops.vorta.* (vorta_amd/torch_ops.py), so
    `torch.compile(fullgraph=True)` traces `HunyuanVideoFlashAttnProcessorTripleEval.__call__` without a graph break
    (aot_eager: fake-tensor tracing through the registered shape functions, no code generation) and the compiled call equals
    the eager one bit for bit -- with the routes dispatched inside the call from the score tensor, and with the device lists
    of a route plan handed in."""
    import vorta_amd
    from vorta_amd import ops
    from vorta_amd.attention import (HunyuanVideoFlashAttnProcessorTripleEval, create_sliding_tile_attn_mask_func,
                                     get_group_info)
    from vorta_amd.routed import HeadRouting
    dtype = torch.bfloat16
    hidden_dim, T, te = 64, 16, 11
    attn = _HyFakeAttn(hidden_dim, dual, dtype, seed=23 + dual)
    torch.manual_seed(9)
    hidden = torch.randn((1, S, hidden_dim), device=dev()).to(dtype)
    enc = torch.randn((1, T, hidden_dim), device=dev()).to(dtype)
    ang = torch.rand((S, 64), device=dev()) * 6.28
    rope = (ang.cos().repeat_interleave(2, dim=1).contiguous(), ang.sin().repeat_interleave(2, dim=1).contiguous())
    mask = torch.zeros((1, 1, 1, S + T), dtype=torch.bool, device=dev())
    mask[..., :S + te] = True
    score = torch.softmax(torch.randn((1, H, 3), device=dev()) * 2, dim=-1)
    kw = dict(lowres_group_info=get_group_info(LATENT, GROUP, 0.5, dev()), window_size=WINDOW, tile_size=TILE,
              latent_shape=LATENT,
              flex_attn_mask_func=create_sliding_tile_attn_mask_func(LATENT, WINDOW, TILE, T, te, dev()))
    proc = HunyuanVideoFlashAttnProcessorTripleEval()
    vorta_amd.set_attention_precision(precision, measurement_only=True)
    try:
        def call(h, e, sc):
            return proc(attn, h, e, mask, rope, routing_score=sc, tau_sparse=0.3, **kw)

        eager = call(hidden, enc, score)
        compiled = torch.compile(call, backend="aot_eager", fullgraph=True)(hidden, enc, score)
        assert torch.equal(eager[0], compiled[0]) and torch.equal(eager[1], compiled[1])
        # the route plan's form: device head lists + counts handed in
        _, lists, counts = ops.route_scores(score, 0.3)

        def call_planned(h, e, sc, lists_, counts_):
            return proc(attn, h, e, mask, rope, routing_score=sc, tau_sparse=0.3,
                        head_routing=HeadRouting.from_device(lists_, counts_), **kw)

        planned = torch.compile(call_planned, backend="aot_eager", fullgraph=True)(hidden, enc, score, lists, counts)
        assert torch.equal(eager[0], planned[0]) and torch.equal(eager[1], planned[1])
    finally:
        vorta_amd.set_attention_precision("native")


def test_wan_processor_call_compiles_fullgraph(golden):
    from vorta_amd.attention import WanAttnProcessorTripleEval
    g = golden("g8_eval_calls")
    dtype = torch.bfloat16
    attn = _WanFakeAttn(g, dtype)
    hidden = torch.tensor(g["wan_hidden"]).to(dtype).to(dev())
    score = torch.tensor(g["routing_score"]).to(dev())
    proc = WanAttnProcessorTripleEval()
    kw = _wan_kwargs()

    def call(h, sc):
        return proc(attn, h, None, None, None, tau_sparse=0.3, routing_score=sc, **kw)

    eager = call(hidden, score)
    compiled = torch.compile(call, backend="aot_eager", fullgraph=True)(hidden, score)
    assert torch.equal(eager, compiled)


@pytest.mark.parametrize("dual", [True, False])
@pytest.mark.parametrize("precision", ["native", "fp8"])
def test_hunyuan_processor_call_is_sync_free_and_graph_capturable(dual, precision):
    """The production call -- projections into one buffer, qk-norm + RoPE, device-resident routes from the score tensor
    (no head counts on the host), fused routed attention, output projection -- enqueues without a host synchronisation
    (torch's sync-debug mode raises on one) and therefore captures into a hipGraph that replays against new activations
    and NEW routes, equal to the eager call bit for bit."""
    import vorta_amd
    from vorta_amd.attention import (HunyuanVideoFlashAttnProcessorTripleEval, create_sliding_tile_attn_mask_func,
                                     get_group_info)
    dtype = torch.bfloat16
    hidden_dim, T, te = 64, 16, 11
    attn = _HyFakeAttn(hidden_dim, dual, dtype, seed=13 + dual)
    torch.manual_seed(8)
    hidden = torch.randn((1, S, hidden_dim), device=dev()).to(dtype)
    enc = torch.randn((1, T, hidden_dim), device=dev()).to(dtype)
    ang = torch.rand((S, 64), device=dev()) * 6.28
    rope = (ang.cos().repeat_interleave(2, dim=1).contiguous(), ang.sin().repeat_interleave(2, dim=1).contiguous())
    mask = torch.zeros((1, 1, 1, S + T), dtype=torch.bool, device=dev())
    mask[..., :S + te] = True
    score = torch.softmax(torch.randn((1, H, 3), device=dev()) * 2, dim=-1)
    kw = dict(lowres_group_info=get_group_info(LATENT, GROUP, 0.5, dev()), window_size=WINDOW, tile_size=TILE,
              latent_shape=LATENT, flex_attn_mask_func=create_sliding_tile_attn_mask_func(LATENT, WINDOW, TILE, T, te, dev()))
    proc = HunyuanVideoFlashAttnProcessorTripleEval()
    vorta_amd.set_attention_precision(precision, measurement_only=True)
    try:
        call = lambda: proc(attn, hidden, enc, mask, rope, routing_score=score, tau_sparse=0.3, **kw)
        call()  # warm-up: geometry tables, allocator pools
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            eager = call()
        finally:
            torch.cuda.set_sync_debug_mode("default")
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            captured = call()
        # new activations and new routes at the captured addresses
        hidden.copy_(torch.randn((1, S, hidden_dim), device=dev()).to(dtype))
        enc.copy_(torch.randn((1, T, hidden_dim), device=dev()).to(dtype))
        score.copy_(torch.softmax(torch.randn((1, H, 3), device=dev()) * 2, dim=-1))
        graph.replay()
        torch.cuda.synchronize()
        replayed = [x.clone() for x in captured]
        again = call()
        assert not torch.equal(replayed[0], eager[0])
        for a, b in zip(replayed, again):
            assert torch.equal(a, b)
    finally:
        vorta_amd.set_attention_precision("native")


@pytest.mark.parametrize("dual", [True, False])
def test_hunyuan_block_projects_into_one_buffer(dual):
    """SURVEY §8f N1 "+ text concat": the video and text projections land in one (1, S+T, H*D) buffer per tensor (the
    GEMMs write it), qk-norm + RoPE run in place on the row ranges -- the same q, k, v as the reference's route
    (separate projections + torch.cat in a dual-stream block, concatenated inputs in a single-stream one), bit for bit,
    and the whole processor call agrees too."""
    from vorta_amd.attention import HunyuanVideoFlashAttnProcessor, hunyuan as hy
    dtype = torch.bfloat16
    hidden_dim, T, te = 64, 16, 11
    attn = _HyFakeAttn(hidden_dim, dual, dtype, seed=9)
    torch.manual_seed(6)
    hidden = torch.randn((1, S, hidden_dim), device=dev()).to(dtype)
    enc = torch.randn((1, T, hidden_dim), device=dev()).to(dtype)
    ang = torch.rand((S, 64), device=dev()) * 6.28
    rope = (ang.cos().repeat_interleave(2, dim=1), ang.sin().repeat_interleave(2, dim=1))
    mask = torch.zeros((1, 1, 1, S + T), dtype=torch.bool, device=dev())
    mask[..., :S + te] = True
    proc = HunyuanVideoFlashAttnProcessor()
    res = {}
    for joint in (True, False):
        hy.JOINT_PROJECTION = joint
        try:
            with torch.no_grad():
                q, k, v, _ = proc._project(attn, hidden, enc, rope)
                out = proc(attn, hidden, enc, mask, rope)
        finally:
            hy.JOINT_PROJECTION = True
        res[joint] = (q, k, v, *out)
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)
    if dual:  # one buffer: (1,H,S+T,D) views of a (1,S+T,H*D) tensor; the reference's route: contiguous concat results
        assert res[True][0].transpose(1, 2).is_contiguous() and res[False][0].is_contiguous()
        # with autograd on and trainable weights the `out=` form is not available: back to the concat route
        q, _, _, _ = proc._project(attn, hidden, enc, rope)
        assert q.is_contiguous() == any(p.requires_grad for p in attn.to_q.parameters())


# --------------------------------------------------------------------------------------------- SP rehearsal
def _sp_worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vorta_amd.attention import WanAttnProcessorTripleEval
    from vorta_amd.ulysses import SP_STATE
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g8_eval_calls.npz"))
    dtype = torch.bfloat16
    attn = _WanFakeAttn(g, dtype)
    hidden = torch.tensor(g["wan_hidden"]).to(dtype).to(dev())
    score = torch.tensor(g["routing_score"]).to(dev())
    proc = WanAttnProcessorTripleEval(check_input=True)
    full = proc(attn, hidden, None, None, None, tau_sparse=0.3, routing_score=score, **_wan_kwargs())
    # a skewed route (one full-attention head, five coreset heads): the placement that follows the routes gives the ranks
    # 2 and 4 heads
    score2 = torch.full_like(score, 0.1)
    for h, e in enumerate([0, 1, 1, 1, 1, 1]):
        score2[0, h, e] = 0.8
    full2 = proc(attn, hidden, None, None, None, tau_sparse=0.3, routing_score=score2, **_wan_kwargs())
    SP_STATE.setup_sp_group(world)
    Sl = S // world
    shard = hidden[:, rank * Sl:(rank + 1) * Sl].contiguous()
    part = proc(attn, shard, None, None, None, tau_sparse=0.3, routing_score=score, **_wan_kwargs())
    err = float((part.float() - full[:, rank * Sl:(rank + 1) * Sl].float()).abs().max().item())
    # ranks holding different numbers of heads (the placement follows the routes), one and two slot groups
    from vorta_amd.attention import _sp
    counts = set()
    for sc, ref in ((score, full), (score2, full2)):
        for groups in (1, 2):
            _sp.SP_PLACEMENT, _sp.SP_GROUPS = "uneven", groups
            _sp._LAYOUTS.clear(); _sp._BUFFERS.clear()
            part = proc(attn, shard, None, None, None, tau_sparse=0.3, routing_score=sc, **_wan_kwargs())
            err = max(err, float((part.float() - ref[:, rank * Sl:(rank + 1) * Sl].float()).abs().max().item()))
            counts |= {k[-1] for k in _sp._LAYOUTS if isinstance(k[-1], tuple)}
    # below whole heads: a full-attention head computes a range of its queries on each rank (VORTA_SP_PLACEMENT=split); the
    # shard must still be the single-process one bit for bit
    splits = 0
    for sc, ref in ((score, full), (score2, full2)):
        for groups in (1, 2):
            _sp.SP_PLACEMENT, _sp.SP_GROUPS = "split", groups
            _sp._LAYOUTS.clear(); _sp._BUFFERS.clear(); _sp._ROUTINGS.clear()
            part = proc(attn, shard, None, None, None, tau_sparse=0.3, routing_score=sc, **_wan_kwargs())
            err = max(err, float((part.float() - ref[:, rank * Sl:(rank + 1) * Sl].float()).abs().max().item()))
            splits += sum(1 for k in _sp._LAYOUTS if isinstance(k[-1], tuple) and sum(k[-1]) > k[0])
    # key splits for under-filled ranks (VORTA_SP_KV_SPLITS=auto; the tiny test geometry always is): the same shard within
    # the 16-bit tolerance (the summation order differs)
    _sp.SP_PLACEMENT, _sp.SP_GROUPS, _sp.SP_KV_SPLITS = "split", 1, "auto"
    _sp._LAYOUTS.clear(); _sp._BUFFERS.clear(); _sp._ROUTINGS.clear()
    part = proc(attn, shard, None, None, None, tau_sparse=0.3, routing_score=score, **_wan_kwargs())
    kerr = float((part.float() - full[:, rank * Sl:(rank + 1) * Sl].float()).abs().max().item())
    _sp.SP_PLACEMENT, _sp.SP_GROUPS, _sp.SP_KV_SPLITS = "uneven", 1, "1"
    ok = (2, 4) in counts and splits > 0 and kerr <= 5e-2  # (unequal head counts seen; a split head seen; key splits close)
    ret[rank] = err if ok else -1.0
    dist.barrier()
    SP_STATE.cleanup()


def test_processor_under_sequence_parallel_rehearsal():
    """2 ranks sharing this GPU (gloo, host-staged messages): the SP branch of the processor returns exactly
    the sequence shard of the single-process result -- with H/P heads on every rank, with head counts that follow the
    routes (`balanced_placement`) and with full-attention heads split by query range (`split_placement`), one and two
    slot groups."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(_sp_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret[0] == 0.0 and ret[1] == 0.0, dict(ret)


def _sp_worker_fp8(rank, world, port, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import vorta_amd
    from vorta_amd import routed
    from vorta_amd.attention import WanAttnProcessorTripleEval
    from vorta_amd.ulysses import SP_STATE
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g8_eval_calls.npz"))
    dtype = torch.bfloat16
    attn = _WanFakeAttn(g, dtype)
    hidden = torch.tensor(g["wan_hidden"]).to(dtype).to(dev())
    score = torch.tensor(g["routing_score"]).to(dev())
    proc = WanAttnProcessorTripleEval(check_input=True)
    native = proc(attn, hidden, None, None, None, tau_sparse=0.3, routing_score=score, **_wan_kwargs())
    from vorta_amd.attention import _sp
    res, full = {}, {}
    Sl = S // world
    shard = hidden[:, rank * Sl:(rank + 1) * Sl].contiguous()
    score2 = torch.full_like(score, 0.1)  # a skewed route: the placement that follows it gives the ranks 2 and 4 heads
    for h, e in enumerate([0, 1, 1, 1, 1, 1]):
        score2[0, h, e] = 0.8
    native2 = proc(attn, hidden, None, None, None, tau_sparse=0.3, routing_score=score2, **_wan_kwargs())
    vorta_amd.set_attention_precision("fp8", measurement_only=True)
    full2 = {}
    for center in (False, True):
        routed.FP8_CENTER_K = center
        full[center] = proc(attn, hidden, None, None, None, tau_sparse=0.3, routing_score=score, **_wan_kwargs())
        full2[center] = proc(attn, hidden, None, None, None, tau_sparse=0.3, routing_score=score2, **_wan_kwargs())
    vorta_amd.set_attention_precision("fp8pv")  # scores in 16 bits, P V in e4m3
    fullpv = proc(attn, hidden, None, None, None, tau_sparse=0.3, routing_score=score, **_wan_kwargs())
    fullpv2 = proc(attn, hidden, None, None, None, tau_sparse=0.3, routing_score=score2, **_wan_kwargs())
    vorta_amd.set_attention_precision("i8pv")  # scores in int8, P V in e4m3
    fulli8 = proc(attn, hidden, None, None, None, tau_sparse=0.3, routing_score=score, **_wan_kwargs())
    fulli8_2 = proc(attn, hidden, None, None, None, tau_sparse=0.3, routing_score=score2, **_wan_kwargs())
    vorta_amd.set_attention_precision("auto8")  # per head int8 or 16-bit scores
    fulla = proc(attn, hidden, None, None, None, tau_sparse=0.3, routing_score=score, **_wan_kwargs())
    assert torch.equal(fulla, fulli8)  # nothing flagged on these inputs
    vorta_amd.set_attention_precision("fp8", measurement_only=True)
    SP_STATE.setup_sp_group(world)
    rel = lambda a, b: float(((a - b) ** 2).mean().sqrt() / (b ** 2).mean().sqrt())
    nat = native[:, rank * Sl:(rank + 1) * Sl].float()
    nat2 = native2[:, rank * Sl:(rank + 1) * Sl].float()
    for groups, v_wire, placement in ((1, True, "even"), (1, False, "even"), (2, True, "even"), (3, False, "even"),
                                      (1, True, "uneven"), (2, False, "uneven")):
        _sp.SP_GROUPS, _sp.SP_V_WIRE, _sp.SP_PLACEMENT = groups, v_wire, placement
        _sp._LAYOUTS.clear(); _sp._BUFFERS.clear()
        for center in (False, True):
            routed.FP8_CENTER_K = center
            sc, fl, nt = (score2, full2, nat2) if placement == "uneven" else (score, full, nat)
            part = proc(attn, shard, None, None, None, tau_sparse=0.3, routing_score=sc, **_wan_kwargs())
            ref = fl[center][:, rank * Sl:(rank + 1) * Sl].float()
            res[(groups, v_wire, placement, center)] = (float((part.float() - ref).abs().max()), rel(ref, nt))
    vorta_amd.set_attention_precision("fp8pv")
    for groups, placement in ((1, "even"), (2, "even"), (2, "uneven")):
        _sp.SP_GROUPS, _sp.SP_V_WIRE, _sp.SP_PLACEMENT = groups, True, placement
        _sp._LAYOUTS.clear(); _sp._BUFFERS.clear()
        sc, fl, nt = (score2, fullpv2, nat2) if placement == "uneven" else (score, fullpv, nat)
        part = proc(attn, shard, None, None, None, tau_sparse=0.3, routing_score=sc, **_wan_kwargs())
        ref = fl[:, rank * Sl:(rank + 1) * Sl].float()
        res[("fp8pv", groups, placement)] = (float((part.float() - ref).abs().max()), rel(ref, nt))
    vorta_amd.set_attention_precision("i8pv")  # k -> int8 on the receive side, slot group by slot group; q in the kernel
    for groups, placement in ((1, "even"), (2, "even"), (2, "uneven")):
        _sp.SP_GROUPS, _sp.SP_V_WIRE, _sp.SP_PLACEMENT = groups, True, placement
        _sp._LAYOUTS.clear(); _sp._BUFFERS.clear()
        sc, fl, nt = (score2, fulli8_2, nat2) if placement == "uneven" else (score, fulli8, nat)
        part = proc(attn, shard, None, None, None, tau_sparse=0.3, routing_score=sc, **_wan_kwargs())
        ref = fl[:, rank * Sl:(rank + 1) * Sl].float()
        res[("i8pv", groups, placement)] = (float((part.float() - ref).abs().max()), rel(ref, nt))
    # "auto8": the per-head choice on the receive layout (tail flags through the row map): the single-process bits
    vorta_amd.set_attention_precision("auto8")
    for groups, placement in ((1, "even"), (2, "even")):
        _sp.SP_GROUPS, _sp.SP_V_WIRE, _sp.SP_PLACEMENT = groups, True, placement
        _sp._LAYOUTS.clear(); _sp._BUFFERS.clear()
        part = proc(attn, shard, None, None, None, tau_sparse=0.3, routing_score=score, **_wan_kwargs())
        ref = fulla[:, rank * Sl:(rank + 1) * Sl].float()
        res[("auto8", groups, placement)] = (float((part.float() - ref).abs().max()), rel(ref, nat))
    vorta_amd.set_attention_precision("native")
    _sp.SP_GROUPS, _sp.SP_V_WIRE, _sp.SP_PLACEMENT = 1, True, "uneven"
    ret[rank] = res
    dist.barrier()
    SP_STATE.cleanup()


def test_processor_under_sequence_parallel_rehearsal_fp8():
    """The same with the e4m3 contractions.  Under SP the receive buffers are converted in the quantiser's segmented row
    layout with the per-head abs-max and the key centre (the mean of the same TOKENS) of the single-process call, so the
    result is bit-identical to it -- with the local heads converted in one go or slot group by slot group (2 + 1 and
    1 + 1 + 1 of the 3 local heads), with v converted on the send side and exchanged as e4m3 or exchanged in 16 bits, and
    with the ranks holding equal or different numbers of heads.  The same for "fp8pv" and for "i8pv" (int8 keys made on the
    receive side per slot group, queries rounded by the kernel's waves: the same waves over the same lists)."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(_sp_worker_fp8, args=(2, port, ret), nprocs=2, join=True)
    for r in (0, 1):
        assert len(ret[r]) == 20
        for key, (d, one) in ret[r].items():
            assert d == 0.0 and 0.0 < one < 0.1, (key, dict(ret[r]))



# --------------------------------------------------------------------------- soft mixture (training forward)
def test_mix_experts_kernel():
    from vorta_amd import ops
    torch.manual_seed(0)
    for dtype in (torch.bfloat16, torch.float16):
        Hh, N = 5, 333
        base = torch.randn((3, N, Hh, 128), device=dev()).to(dtype)   # strided (H,N,D) views, like the processors'
        xs = [base[e].permute(1, 0, 2) for e in range(3)]
        sc = torch.softmax(torch.randn((1, Hh, 3), device=dev()), -1).to(dtype)
        out = torch.empty((N, Hh, 128), device=dev(), dtype=dtype).permute(1, 0, 2)
        ops.mix_experts(xs, sc, out)
        want = sum(sc[0, :, e, None, None].double() * xs[e].double() for e in range(3))
        # one rounding of an fp32 sum: half an ulp of the result
        assert (out.double() - want).abs().max() <= (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11) * want.abs().max()
        ops.mix_experts(xs, sc, xs[1])  # the output may alias an input
        assert torch.equal(xs[1], out)
    with pytest.raises(ValueError):
        ops.mix_experts(xs[:2], sc, out)


def test_wan_soft_mixture_whole_call_golden(golden):
    """WanAttnProcessorTripleTrain.__call__ forward (wan.py:195-241) against the reference's output (G11)."""
    from vorta_amd.attention import WanAttnProcessorTripleTrain
    g8, g = golden("g8_eval_calls"), golden("g11_soft_mixture")
    dtype = torch.bfloat16
    attn = _WanFakeAttn(g8, dtype)
    hidden = torch.tensor(g8["wan_hidden"]).to(dtype).to(dev())
    proc = WanAttnProcessorTripleTrain(check_input=True)
    kw = {k: v for k, v in _wan_kwargs().items() if k != "tau_sparse"}
    sc = torch.tensor(g["routing_score"]).to(dev())
    with torch.no_grad():
        y = proc(attn, hidden, None, None, None, routing_score=sc, **kw)
        teacher = proc(attn, hidden, None, None, None, use_original_attn=True)
    for got, gold in ((y, g["wan_soft_out"]), (teacher, g["wan_teacher_out"])):
        assert got.shape == gold.shape
        assert rel_fro(got.float().cpu().numpy(), gold) < 2e-2
    # gradients are refused loudly, not dropped
    with pytest.raises(NotImplementedError):
        proc(attn, hidden.clone().requires_grad_(True), None, None, None, routing_score=sc, **kw)


def test_hunyuan_soft_mixture_vs_oracle_and_golden(golden):
    from vorta_amd.routed import geometry_for, soft_mixture_attention
    g8, g = golden("g8_eval_calls"), golden("g11_soft_mixture")
    t, te = (int(x) for x in g8["text"])
    geom = geometry_for(LATENT, TILE, WINDOW, GROUP, 0.5, dev())
    gold = np.concatenate([g["hy_soft_out"], g["hy_soft_eout"]], axis=2)
    for dtype in (torch.bfloat16, torch.float16):
        # golden q,k,v have head dim 16: zero-pad to 128 (scores and outputs unchanged), scale passed explicitly
        def pad(x):
            out = torch.zeros(x.shape[:-1] + (128,), dtype=dtype, device=dev())
            out[..., :16] = torch.tensor(x).to(dtype).to(dev())
            return out
        q, k, v = pad(g8["hy_q"]), pad(g8["hy_k"]), pad(g8["hy_v"])
        sc = torch.tensor(g["routing_score"]).to(dev())
        out = soft_mixture_attention(q, k, v, sc, geom, model="hunyuan", text_len=t, text_valid=te, scale=1.0 / 4.0)
        ref = O.soft_mixture_attention(q[..., :16].double().cpu().numpy(), k[..., :16].double().cpu().numpy(),
                                       v[..., :16].double().cpu().numpy(), g["routing_score"], model="hunyuan",
                                       latent=LATENT, tile=TILE, window=WINDOW, gi=O.group_info(LATENT, GROUP, 0.5),
                                       t_text=t, t_eff=te)
        got = out[..., :16].float().cpu().numpy()
        assert rel_fro(got, ref) < 1e-2  # same (rounded) inputs: same coreset rankings
        assert (got[:, :, S + te:] == 0).all()
        if dtype == torch.float16:
            # vs the reference's own fp32 run: with bf16 inputs the rounding reorders a few coreset rankings (every
            # head runs the coreset expert here), so the golden comparison is made in fp16
            assert rel_fro(got, gold) < 2e-2


# --------------------------------------------------------------------------- Wan cross attention with image tokens
class _WanI2VAttn(nn.Module):
    """cross-attention module of the image-to-video models: 257 image tokens in front of the text (wan.py:70-73)"""

    def __init__(self, dim, dtype, seed=21):
        super().__init__()
        torch.manual_seed(seed)
        inner = 4 * 128
        self.heads = 4
        self.to_q, self.to_k, self.to_v = nn.Linear(dim, inner), nn.Linear(dim, inner), nn.Linear(dim, inner)
        self.add_k_proj, self.add_v_proj = nn.Linear(dim, inner), nn.Linear(dim, inner)
        self.norm_q, self.norm_k, self.norm_added_k = (nn.RMSNorm(inner, eps=1e-6) for _ in range(3))
        for m in (self.norm_q, self.norm_k, self.norm_added_k):
            nn.init.uniform_(m.weight, 0.5, 1.5)
        self.to_out = nn.ModuleList([nn.Linear(inner, dim), nn.Identity()])
        self.to(dev()).to(dtype)


def test_wan_cross_attention_with_image_tokens_vs_oracle():
    """wan.py:70-73,121-139: the first 257 encoder tokens are image tokens with their own K/V projections; their
    attention output is added to the text cross-attention output before the output projection."""
    from vorta_amd.attention import WanAttnProcessor2_0
    dtype = torch.bfloat16
    dim, Sq, n_txt = 96, 300, 40
    attn = _WanI2VAttn(dim, dtype)
    torch.manual_seed(22)
    hidden = torch.randn((2, Sq, dim), device=dev()).to(dtype)  # batch 2: the dense processor serves any batch
    enc = torch.randn((2, 257 + n_txt, dim), device=dev()).to(dtype)
    y = WanAttnProcessor2_0()(attn, hidden, enc, None, None)

    def lin(m, x):
        return x @ _f64(m.weight).T + _f64(m.bias)

    def rms(m, x):
        return x / np.sqrt((x * x).mean(-1, keepdims=True) + 1e-6) * _f64(m.weight)

    def heads(x):
        return x.reshape(x.shape[0], x.shape[1], 4, 128).transpose(0, 2, 1, 3)

    h, e = _f64(hidden), _f64(enc)
    img, txt = e[:, :257], e[:, 257:]
    q = heads(rms(attn.norm_q, lin(attn.to_q, h)))
    k, v = heads(rms(attn.norm_k, lin(attn.to_k, txt))), heads(lin(attn.to_v, txt))
    ki, vi = heads(rms(attn.norm_added_k, lin(attn.add_k_proj, img))), heads(lin(attn.add_v_proj, img))
    o = O.dense_attention(q, k, v) + O.dense_attention(q, ki, vi)
    want = lin(attn.to_out[0], o.transpose(0, 2, 1, 3).reshape(2, Sq, 512))
    assert y.shape == want.shape
    assert rel_fro(y.float().cpu().numpy(), want) < 1.5e-2
