#!/usr/bin/env python
"""bench.py -- denoising-step attention time of VORTA's routed sparse attention on MI355X.

One "step" = one denoising step of the named model = every attention layer of the transformer once
(x forwards per step), each layer being ONE routed-attention op (router dispatch is given, experts:
full / coreset / sliding tile) on synthetic post-RoPE Q/K/V already resident in HBM.

    python bench.py [--gpus N --steps K --warmup W] [--config hunyuan-129f] [--mix uniform]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N>1: Ulysses over RCCL)

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` and, at N=1,
`cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

# ------------------------------------------------------------------------------------------------ configs
# latent = pixel2token(video) (vorta/patch/utils.py:76-92); H/layers/text from the public model configs
# (SURVEY.md §8); tile/window/group: SURVEY.md §8(d) table ("proposed" = our stated choice where the
# authors' geometry does not divide the latent grid; the authors' own shapes are listed too).
CONFIGS = {
    # BASELINE.json configs[2]: the configuration the metric is quoted on
    "hunyuan-129f": dict(model="hunyuan", latent=(33, 45, 80), heads=24, layers=60, fwd_per_step=1, text=256,
                         text_valid=96, tile=(11, 9, 8), window=(3, 3, 3), group=(3, 3, 2), rate=0.5, dtype="fp16"),
    # authors' own benchmark shape (vorta/constants.py:8-12, scripts/hunyuan/train.sh:10-18)
    "hunyuan-117f": dict(model="hunyuan", latent=(30, 45, 80), heads=24, layers=60, fwd_per_step=1, text=256,
                         text_valid=96, tile=(6, 9, 8), window=(3, 3, 3), group=(2, 3, 2), rate=0.5, dtype="bf16"),
    # BASELINE.json configs[0]: Wan-2.1 1.3B 49x320x512, --native_attention (scripts/wan/inference.py:154-163 -> wan.py:142-145):
    # every head takes the dense expert (the mix is forced to all-full); its CPU leg runs IN FULL
    "wan1.3b-49f": dict(model="wan", latent=(13, 20, 32), heads=12, layers=30, fwd_per_step=2, text=0, text_valid=0,
                        tile=(13, 10, 8), window=(3, 3, 3), group=(1, 2, 2), rate=0.5, dtype="bf16", native_only=True),
    # BASELINE.json configs[1]
    "wan1.3b-81f": dict(model="wan", latent=(21, 30, 52), heads=12, layers=30, fwd_per_step=2, text=0, text_valid=0,
                        tile=(7, 6, 4), window=(3, 3, 3), group=(3, 3, 2), rate=0.5, dtype="bf16"),
    # BASELINE.json configs[4] geometry; --dtype fp8 runs its fp8 MFMA path (bf16 is the same-box A/B arm)
    "wan14b-81f": dict(model="wan", latent=(21, 45, 80), heads=40, layers=40, fwd_per_step=2, text=0, text_valid=0,
                       tile=(7, 9, 8), window=(3, 3, 3), group=(3, 3, 2), rate=0.5, dtype="bf16"),
    "wan14b-77f": dict(model="wan", latent=(20, 45, 80), heads=40, layers=40, fwd_per_step=2, text=0, text_valid=0,
                       tile=(5, 9, 8), window=(3, 3, 3), group=(2, 3, 2), rate=0.5, dtype="bf16"),
    "wan-tiny": dict(model="wan", latent=(9, 12, 16), heads=8, layers=4, fwd_per_step=2, text=0, text_valid=0,
                     tile=(3, 6, 8), window=(3, 3, 3), group=(3, 3, 2), rate=0.5, dtype="bf16"),
    # tiny, for rehearsals
    "tiny": dict(model="hunyuan", latent=(9, 12, 16), heads=8, layers=4, fwd_per_step=1, text=64, text_valid=40,
                 tile=(3, 6, 8), window=(3, 3, 3), group=(3, 3, 2), rate=0.5, dtype="bf16"),
}
# routing mixes (fractions full / lowres / sliding), SURVEY.md §8(d): no router weights exist offline
MIXES = {"all-full": (1, 0, 0), "all-lowres": (0, 1, 0), "all-sliding": (0, 0, 1), "uniform": (1 / 3, 1 / 3, 1 / 3),
         "sparse-heavy": (1 / 6, 1 / 3, 1 / 2)}

PEAK_MFMA_TFLOPS = 2500.0  # dense bf16/fp16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_MFMA_FP8_TFLOPS = 5000.0  # dense fp8 MFMA peak (same table; the 10 PF figure includes 2:1 sparsity)


def layer_experts(cfg, mix, layer, heads=None):
    """expert id per head for one layer: counts fixed by the mix, assignment drawn with rng(1234+layer)."""
    H = cfg["heads"] if heads is None else heads
    f = MIXES[mix]
    n1 = int(round(H * f[1]))
    n2 = int(round(H * f[2]))
    n0 = H - n1 - n2
    ids = np.array([0] * n0 + [1] * n1 + [2] * n2)
    return np.random.default_rng(1234 + layer).permutation(ids)


def algorithmic_flops(cfg, experts):
    """SURVEY.md §8(d) / BASELINE.md §2 per-layer work for a given head->expert assignment."""
    from vorta_amd import ops
    S = cfg["latent"][0] * cfg["latent"][1] * cfg["latent"][2]
    te, D = cfg["text_valid"], 128
    g = cfg["group"][0] * cfg["group"][1] * cfg["group"][2]
    s_low = (S // g) * (1 + int(g * (1 - cfg["rate"])) - 1)
    _, tok, n_kv = ops.sta_table_sizes(cfg["latent"], cfg["tile"], cfg["window"], te)
    f_full = 4.0 * (S + te) ** 2 * D
    f_low = 4.0 * (s_low + te) ** 2 * D
    f_sl = 4.0 * D * (S * n_kv + te * (S + te))
    n = [int((experts == e).sum()) for e in range(3)]
    return n[0] * f_full + n[1] * f_low + n[2] * f_sl, dict(full=f_full, lowres=f_low, sliding=f_sl)


def cpu_baseline_full(cfg, cores, cpu_model, gpu_dtype, steps: int = 2):
    """BASELINE.json configs[0] / BASELINE.md section 3: the --native_attention path of Wan-2.1 1.3B at 49x320x512 on the CPU, run
    IN FULL -- `steps` denoising steps x 2 forwards (CFG) x 30 layers, every layer the dense attention of all 12 heads on its own
    seeded post-RoPE q, k, v (torch-CPU SDPA, the call the reference makes at wan.py:142-145), nothing extrapolated."""
    import torch.nn.functional as F
    H, L, fwd = cfg["heads"], cfg["layers"], cfg["fwd_per_step"]
    S = cfg["latent"][0] * cfg["latent"][1] * cfg["latent"][2]
    dt = torch.bfloat16
    sets = []
    for i in range(2):
        gen = torch.Generator().manual_seed(1234 + i)
        sets.append(tuple(torch.randn((1, H, S, 128), generator=gen).to(dt) for _ in range(3)))
    with torch.no_grad():
        F.scaled_dot_product_attention(*sets[0])  # warm the dispatcher and the thread pool
        t0 = time.perf_counter()
        for _ in range(steps * fwd):
            for l in range(L):
                F.scaled_dot_product_attention(*sets[l % 2])
        step_s = (time.perf_counter() - t0) / steps
    return {"value": S * fwd / step_s, "unit": "video_tokens/s", "cores": int(cores), "kind": "full", "cpu_model": cpu_model,
            "dtype": "bf16", "gpu_dtype": gpu_dtype, "step_s": round(step_s, 2),
            "sample": f"the whole workload, nothing extrapolated: {steps} denoising steps x {fwd} forwards x {L} layers of dense "
                      f"attention over {H} heads x S = {S} (torch-CPU SDPA, bf16, {cores} threads); the port of the path "
                      "(oracle-checked torch-CPU restatement), not the reference's own Python"}


def cpu_baseline(cfg, layer_ids, gpu_dtype="bf16"):
    """SURVEY.md §8(d) / BASELINE.md §3: the CPU restatement of the path timed on this host -- torch-CPU SDPA for the
    contractions, the oracle's index code (oracle/vorta_oracle.py: group tables + cosine ranking, tile-major order,
    clamped tile windows) for the coreset selection / pooling / unpooling and the sliding-tile key lists -- ONE head per
    expert at the full sequence length of the configuration (a bounded sample: seconds), extrapolated by the number of
    heads per expert, layers and forwards per step.  The reference's own Python cannot travel to the GPU box."""
    import torch.nn.functional as F
    from oracle import vorta_oracle as O
    cores = min(os.cpu_count() or 1, 64)  # beyond one socket's cores SDPA on one head slows down
    torch.set_num_threads(cores)
    cpu_model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    latent, tile, window, group = cfg["latent"], cfg["tile"], cfg["window"], cfg["group"]
    S = latent[0] * latent[1] * latent[2]
    hy = cfg["model"] == "hunyuan"
    T, te = (cfg["text"], cfg["text_valid"]) if hy else (0, 0)
    dt = torch.bfloat16
    if cfg.get("native_only"):
        return cpu_baseline_full(cfg, cores, cpu_model, gpu_dtype)
    gen = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn((1, 1, S + T, 128), generator=gen).to(dt) for _ in range(3))
    sdpa = lambda a, b, c: F.scaled_dot_product_attention(a, b, c)
    times = {}
    with torch.no_grad():
        sdpa(q[:, :, :256], k[:, :, :256], v[:, :, :256])  # warm the dispatcher
        # expert 0: dense on the valid keys (hunyuan.py:167-176)
        t0 = time.perf_counter()
        sdpa(q[:, :, :S + te], k[:, :, :S + te], v[:, :, :S + te])
        times["full"] = time.perf_counter() - t0
        # expert 1: rank, pool, dense on the packed sequence, unpool (coreset_select.py:68-185)
        t0 = time.perf_counter()
        gi = O.group_info(latent, group, cfg["rate"])
        qv, kv = q[:, :, :S].float().numpy(), k[:, :, :S].float().numpy()
        mq = O.coreset_match(qv, gi)
        mk = O.coreset_match(kv, gi) if hy else mq
        keep_q, drop_q = O.coreset_row_lists(gi, *mq)
        keep_k, _ = O.coreset_row_lists(gi, *mk)
        iq = torch.cat([torch.from_numpy(keep_q[0, 0]), torch.arange(S, S + te)])
        ik = torch.cat([torch.from_numpy(keep_k[0, 0]), torch.arange(S, S + te)])
        o_low = sdpa(q[:, :, iq], k[:, :, ik], v[:, :, ik])
        out = torch.zeros_like(q)
        out[:, :, iq] = o_low
        centres = o_low[:, :, :gi.n_groups]
        out[:, :, torch.from_numpy(drop_q[0, 0]).reshape(-1)] = centres.repeat_interleave(drop_q.shape[-1], dim=2)
        times["lowres"] = time.perf_counter() - t0
        # expert 2: tile-major order, per query tile the keys of its clamped window (+ valid text), text queries see all
        t0 = time.perf_counter()
        perm = torch.from_numpy(O.tile_major_order(latent, tile))
        sees = O.sta_window_tiles(latent, tile, window)
        tok = tile[0] * tile[1] * tile[2]
        text_keys = torch.arange(S, S + te)
        n_tiles = S // tok
        sample = sorted(set(int(x) for x in np.linspace(0, n_tiles - 1, min(n_tiles, 16))))  # bounded: 16 evenly spaced tiles
        for ti in sample:
            keys = torch.cat([perm[j * tok:(j + 1) * tok] for j in np.nonzero(sees[ti])[0]] + [text_keys])
            rows = perm[ti * tok:(ti + 1) * tok]
            out[:, :, rows] = sdpa(q[:, :, rows], k[:, :, keys], v[:, :, keys])
        t_tiles = (time.perf_counter() - t0) * n_tiles / len(sample)
        t0 = time.perf_counter()
        if te:
            out[:, :, S:S + te] = sdpa(q[:, :, S:S + te], k[:, :, :S + te], v[:, :, :S + te])
        times["sliding"] = t_tiles + (time.perf_counter() - t0)
    layer_s = 0.0
    for e in layer_ids:
        n = [int((e == i).sum()) for i in range(3)]
        layer_s += n[0] * times["full"] + n[1] * times["lowres"] + n[2] * times["sliding"]
    step_s = layer_s * cfg["fwd_per_step"]
    tokens = S * cfg["fwd_per_step"]
    return {"value": tokens / step_s, "unit": "video_tokens/s", "cores": int(cores), "kind": "port", "cpu_model": cpu_model,
            "dtype": "bf16", "gpu_dtype": gpu_dtype, "step_s_extrapolated": round(step_s, 1),
            "seconds_per_head": {k_: round(v_, 3) for k_, v_ in times.items()},
            "sample": f"one head per expert at the full sequence (S = {S}, text {T}/{te}): torch-CPU SDPA (bf16, "
                      f"{cores} threads) + the oracle's index code for coreset ranking / pooling / unpooling and the "
                      f"sliding-tile key lists ({len(sample)} of its {n_tiles} query tiles timed); extrapolated x heads per expert x {cfg['layers']} layers x "
                      f"{cfg['fwd_per_step']} forward(s) per step"}


def processor_level(cfg, mix, dev, dt, fp8):
    """The production call path of one denoising step (hunyuan.py:521-610 / wan.py:308-386 behind
    modeling_hunyuan.py:491-499,555-563): modules of the model's real width with random weights (the shapes of
    diffusers' Attention: to_q/k/v, norm_q/k, add_*_proj + norm_added_* + to_add_out in a dual-stream block, to_out),
    one Router per block whose bias encodes the layer's synthetic route (weights 0: softmax(bias) puts 0.999 on the
    chosen expert), the step's routes from ONE RoutePlan.compute inside the timed step, every layer's processor called
    the way BoundProcessor calls it (patch/_engine.py:148-154: scores + device head lists of the plan).
    Returns (one_step, info)."""
    from torch import nn

    import vorta_amd
    from vorta_amd.attention import (HunyuanVideoFlashAttnProcessorTripleEval, WanAttnProcessorTripleEval,
                                     create_sliding_tile_attn_mask_func)
    from vorta_amd.patch._engine import RoutePlan
    from vorta_amd.patch.router import Router
    from vorta_amd.patch.utils import prepare_hunyuan_self_attn_kwargs, prepare_wan_self_attn_kwargs
    H, L, T, te = cfg["heads"], cfg["layers"], cfg["text"], cfg["text_valid"]
    S = cfg["latent"][0] * cfg["latent"][1] * cfg["latent"][2]
    hy = cfg["model"] == "hunyuan"
    width = H * 128
    vorta_amd.set_attention_precision(fp8 if fp8 in ("fp8pv", "i8pv", "auto8") else "fp8" if fp8 else "native",
                                      measurement_only=True)
    gen = torch.Generator(device=dev).manual_seed(1234)

    def lin(i, o):
        m = nn.Linear(i, o, bias=True, device=dev, dtype=dt)
        with torch.no_grad():
            m.weight.normal_(0.0, i ** -0.5, generator=gen)
            m.bias.normal_(0.0, 0.02, generator=gen)
        return m.requires_grad_(False)

    def rms(n):
        return nn.RMSNorm(n, eps=1e-6, device=dev, dtype=dt).requires_grad_(False)

    class Attn(nn.Module):
        def __init__(self, dual):
            super().__init__()
            self.heads = H
            self.to_q, self.to_k, self.to_v = lin(width, width), lin(width, width), lin(width, width)
            self.norm_q, self.norm_k = (rms(128), rms(128)) if hy else (rms(width), rms(width))
            self.add_q_proj = self.add_k_proj = self.add_v_proj = None
            self.norm_added_q = self.norm_added_k = self.to_add_out = None
            if dual:
                self.add_q_proj, self.add_k_proj, self.add_v_proj = lin(width, width), lin(width, width), lin(width, width)
                self.norm_added_q, self.norm_added_k = rms(128), rms(128)
                self.to_add_out = lin(width, width)
            # single-stream Hunyuan blocks are `pre_only` (no to_out: proj_out of the block follows); Wan always projects
            self.to_out = nn.ModuleList([lin(width, width), nn.Dropout(0.0)]) if (dual or not hy) else None

    # HunyuanVideo: 20 dual-stream + 40 single-stream blocks (public config); Wan: every block alike
    n_dual = L // 3 if hy else 0
    dual_attn, single_attn = (Attn(True) if n_dual else None), Attn(False)
    layer_ids = [layer_experts(cfg, mix, l) for l in range(L)]
    routers = []
    for e in layer_ids:
        r = Router(width, H).to(device=dev, dtype=torch.bfloat16)
        with torch.no_grad():
            r.linear.weight.zero_()
            b = torch.zeros(H, 3)
            b[torch.arange(H), torch.as_tensor(e)] = 8.0
            r.linear.bias.copy_(b.reshape(-1))
        routers.append(r)
    plan = RoutePlan(routers)
    temb = torch.randn((1, width), generator=gen, device=dev, dtype=dt)
    hidden = torch.randn((1, S, width), generator=gen, device=dev, dtype=dt)
    tau = 0.5
    sak = dict(latent_shape=cfg["latent"], window_size=cfg["window"], tile_size=cfg["tile"],
               lowres_window_size=cfg["group"], lowres_reduction_rate=cfg["rate"])
    if hy:
        enc = torch.randn((1, T, width), generator=gen, device=dev, dtype=dt)
        mask = torch.zeros((1, 1, 1, S + T), dtype=torch.bool, device=dev)
        mask[..., :S + te] = True
        ang = torch.rand((S, 64), generator=gen, device=dev) * 6.28
        rope = (ang.cos().repeat_interleave(2, dim=1).contiguous(), ang.sin().repeat_interleave(2, dim=1).contiguous())
        kw = prepare_hunyuan_self_attn_kwargs(sak, dev, tau_sparse=tau)
        # per prompt, outside the step (pipeline_hunyuan.py:378-392)
        kw["flex_attn_mask_func"] = create_sliding_tile_attn_mask_func(cfg["latent"], cfg["window"], cfg["tile"], T, te, dev)
        proc = HunyuanVideoFlashAttnProcessorTripleEval()
    else:
        ang = torch.rand((1, 1, S, 64), generator=gen, device=dev, dtype=torch.float64) * 6.28
        rope = torch.polar(torch.ones_like(ang), ang)
        kw = prepare_wan_self_attn_kwargs(sak, dev, tau_sparse=tau)
        proc = WanAttnProcessorTripleEval()
    sink = []

    def one_step():
        plan.compute(temb, tau)
        for _ in range(cfg["fwd_per_step"]):
            for l in range(L):
                if hy:
                    a = dual_attn if l < n_dual else single_attn
                    out = proc(a, hidden, enc, mask, rope, routing_score=plan.scores(l), head_routing=plan.routing(l), **kw)
                else:
                    out = proc(single_attn, hidden, None, None, rope, routing_score=plan.scores(l),
                               head_routing=plan.routing(l), **kw)
        sink[:] = [out]

    def fingerprint_tensor():
        o = sink[0]
        return o[0] if isinstance(o, tuple) else o

    # the plan must reproduce the synthetic routes exactly (checked once, outside the timed region)
    plan.compute(temb, tau)
    got = plan._out[1].cpu().numpy()
    assert (got == np.stack(layer_ids)).all(), "route plan does not reproduce the synthetic routing mix"
    info = {"level": "processor", "width": width, "blocks": (f"{n_dual} dual-stream + {L - n_dual} single-stream" if hy
                                                            else f"{L} self-attention"),
            "routes": "device-resident (RoutePlan.routing: head lists + counts, every expert grid sized for H slots)",
            "sync_debug_mode": "error"}
    return one_step, fingerprint_tensor, layer_ids, info


XGMI_LINK_GBPS = 153.0  # per direction and link, 7 links per GPU (MI355X_MICROARCH.md)


def exchange_breakdown(sp, cfg, args, ms_per_step, barrier, dist, dev, world, fp8, backend,
                       reps: int = int(os.environ.get("VORTA_BENCH_BREAKDOWN_REPS", "2"))):
    """Two more passes over the same layers, each bracketed like the timed region (barrier + synchronize, MAX over ranks):
      * exchange only -- staging pass, every slot group's all_to_all_single in and back, text all-gather, un-permute, NO attention:
        `exchange_ms_per_layer`, and with the bytes a rank puts on each link, the rate the links reached;
      * compute only -- the same layers with the layouts in loopback (every local pass, no transfer): `compute_ms_per_layer`.
    `exposed_exchange_ms_per_layer` = timed step / layers - compute only: what of the exchange the step did NOT hide behind
    attention (with one slot group: all of it; with several: what is left).  Bytes are counted from the layouts: per layer and
    tensor a rank sends `heads of peer j x S/P x D` elements to peer j over their link."""
    L, fwd = cfg["layers"], cfg["fwd_per_step"]
    n_layers = L * fwd

    def timed(fn):
        fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        barrier()
        t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) * 1e3 / reps / n_layers

    def all_layers(**kw):
        for _ in range(fwd):
            for l in range(L):
                sp.layer(l, **kw)

    ex_ms = timed(lambda: all_layers(exchange_only=True))
    lays = list({id(x): x for x in sp.lays}.values())
    for x in lays:
        x.loopback = True
    try:
        comp_ms = timed(all_layers)
    finally:
        for x in lays:
            x.loopback = False
    esz = 2
    v_bytes = 1 if (fp8 and not args.no_v_wire) or fp8 in ("fp8pv", "i8pv", "auto8") else esz
    link_in = link_out = egress = 0.0
    for lay in sp.lays:
        peers = [j for j in range(lay.P) if j != lay.rank]
        chunk = lambda heads: heads * lay.Sl * lay.D
        if not peers:  # a world of one (the RCCL rehearsal): nothing leaves the rank
            continue
        link_in += max(chunk(lay.counts[j]) for j in peers) * (2 * esz + v_bytes)  # q, k, v to the peer holding most heads
        link_out += chunk(lay.Hl) * esz  # o back: this rank's slots to every peer
        egress += sum(chunk(lay.counts[j]) for j in peers) * (2 * esz + v_bytes) + len(peers) * chunk(lay.Hl) * esz
    link_bytes = (link_in + link_out) / len(sp.lays)
    rate = link_bytes / (ex_ms * 1e-3) / 1e9 if ex_ms > 0 else 0.0
    step_ms_per_layer = ms_per_step / n_layers
    return {"exchange_ms_per_layer": round(ex_ms, 4), "compute_ms_per_layer": round(comp_ms, 4),
            "step_ms_per_layer": round(step_ms_per_layer, 4),
            "exposed_exchange_ms_per_layer": round(max(step_ms_per_layer - comp_ms, 0.0), 4),
            "hidden_fraction_of_exchange": round(1.0 - max(step_ms_per_layer - comp_ms, 0.0) / ex_ms, 3) if ex_ms > 0 else None,
            "bytes_per_link_per_layer": int(link_bytes), "bytes_per_link_per_16bit_tensor": int(link_bytes * esz / (3 * esz + v_bytes)),
            "egress_bytes_per_rank_per_layer": int(egress / len(sp.lays)), "v_bytes_per_element_on_the_wire": v_bytes,
            "link_GBps_in_exchange_only": round(rate, 1), "frac_of_link_peak": round(rate / XGMI_LINK_GBPS, 3),
            "aggregate_egress_GBps": round(egress / len(sp.lays) / (ex_ms * 1e-3) / 1e9, 1) if ex_ms > 0 else 0.0,
            "frac_of_7_links": round(egress / len(sp.lays) / (ex_ms * 1e-3) / 1e9 / (7 * XGMI_LINK_GBPS), 3) if ex_ms > 0 else 0.0,
            "link_peak_GBps": XGMI_LINK_GBPS, "slot_groups": args.sp_groups,
            "what": f"rank 0's layouts, mean over {len(sp.lays)} layers; times = MAX over ranks of {reps} passes, barrier-bracketed; "
                    "exchange only = staging + all_to_all_single in and back + text all-gather, no attention; compute only = the "
                    "layers in loopback (no transfer)" + ("; gloo rehearsal: host-staged messages, the rates say nothing about xGMI"
                                                           if backend != "nccl" else "")}


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py
    <same arguments>` as a child (one rank per GPU, rendezvous on 127.0.0.1 at a free port) and pass its stdout / stderr
    and exit code through.  Called before anything initialises HIP in this process."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] launching {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def supervise(args):
    """One of the N processes a launcher (torch.distributed.run) started for a multi-GPU run, acting as a GPU-FREE supervisor: it
    runs the measurement in a CHILD process (same arguments, same rank environment; never an exec) and, when ANY rank's first
    attempt fails -- a self-check mismatch, an RCCL error, a collective that never completes -- starts ONE fresh child per rank
    with `--conservative` (auto placement, one slot group, v in 16 bits: the oldest form of the exchange), whose line carries
    `"fallback": "conservative"` and `"first_attempt_error"`.  The supervisors agree through the launcher's own TCPStore
    (TORCHELASTIC_USE_AGENT_STORE; each attempt's ranks rendezvous under their own prefix of it, `attempt_store`), so a
    failed first attempt cannot leave a multi-GPU run without a number.  Rank 0's supervisor relays its child's stdout: JSON lines to
    stdout, everything else (RCCL's version banner, warnings) to stderr; the JSON lines of a failed attempt that is followed by a
    fallback go to stderr too, so stdout carries ONE JSON line and nothing else."""
    import subprocess
    from datetime import timedelta

    from torch.distributed import PrefixStore, TCPStore
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    try:
        store = PrefixStore("/vorta_bench_supervisor", TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]), world,
                                                                False, timedelta(seconds=120)))
        store.set(f"hello/{rank}", "1")
        store.wait([f"hello/{j}" for j in range(world)], timedelta(seconds=120))  # every rank supervises, or none does
    except Exception as exc:  # noqa: BLE001 -- no store to agree through: this process runs the measurement itself, as before round 6
        print(f"[bench] rank {rank}: cannot reach the launcher's store ({type(exc).__name__}: {exc}); running unsupervised",
              file=sys.stderr, flush=True)
        return None
    cap_s = float(os.environ.get("VORTA_BENCH_ATTEMPT_TIMEOUT_S", "1500"))
    current = {}

    def on_term(signum, frame):  # the launcher ends its processes: end exactly the child this supervisor started, then leave
        p_ = current.get("proc")
        if p_ is not None and p_.poll() is None:
            p_.kill()
        os._exit(128 + signum)
    import signal
    signal.signal(signal.SIGTERM, on_term)

    def attempt(n: int, extra, first_error):
        env = dict(os.environ, VORTA_BENCH_WORKER="1", VORTA_BENCH_ATTEMPT=str(n))  # (`attempt_store`: a namespace per attempt)
        if first_error is not None:
            env["VORTA_BENCH_FIRST_ATTEMPT_ERROR"] = first_error[:600]
        cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + extra
        # (the other ranks' children print nothing of the contract: their stdout joins stderr, stdout stays the JSON line's)
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if rank == 0 else sys.stderr, text=True)
        current["proc"] = proc
        held = []
        if rank == 0:
            import threading

            def pump():
                for line in proc.stdout:
                    if line.startswith("{"):
                        held.append(line.rstrip("\n"))
                    else:  # (RCCL prints its version banner on stdout: stdout is kept for the JSON line)
                        sys.stderr.write(line)
                        sys.stderr.flush()
            th = threading.Thread(target=pump, daemon=True)
            th.start()
        t0 = time.perf_counter()
        killed = False
        while proc.poll() is None:
            time.sleep(0.5)
            # another rank's attempt has failed: this rank's child can only be waiting for it -- give it a moment to leave by
            # itself (its process group's watchdog), then end exactly this child
            if not killed and (store.check([f"a{n}/failed"]) and time.perf_counter() - float(store.get(f"a{n}/failed")) > 20.0
                               or time.perf_counter() - t0 > cap_s):
                proc.kill()
                killed = True
        rc = proc.returncode
        if rank == 0:
            th.join(10.0)
        if rc != 0 and not store.check([f"a{n}/failed"]):
            store.set(f"a{n}/failed", repr(time.perf_counter()))  # (one host: the supervisors share a clock)
        store.set(f"a{n}/rc/{rank}", str(rc))
        store.wait([f"a{n}/rc/{j}" for j in range(world)], timedelta(seconds=cap_s + 120))
        rcs = [int(store.get(f"a{n}/rc/{j}")) for j in range(world)]
        return rc, rcs, held

    rc, rcs, held = attempt(0, [], None)
    # Rank 0 decides for everyone (the others cannot see its child's stdout): the attempt counts when every child left with 0 --
    # or when the measurement was made and printed and only the TEARDOWN of some rank failed (a crash in a communicator's
    # destructor must not throw a finished measurement away).  A failed attempt is final under --conservative / --no-fallback.
    if rank == 0:
        measured = any(l.startswith('{"metric"') for l in held)
        clean = all(x == 0 for x in rcs)
        first_error = f"exit codes {rcs}"
        errs = [l for l in held if l.startswith('{"error"')]
        if errs:
            try:
                first_error = json.loads(errs[-1]).get("error", first_error) + f" (exit codes {rcs})"
            except ValueError:
                pass
        if clean or measured or args.conservative or args.no_fallback:
            decision = "accept" if (clean or measured) else "final"
        else:
            decision = "fallback"
        store.set("a0/decision", decision + "\n" + first_error)
    else:
        store.wait(["a0/decision"], timedelta(seconds=120))
    decision, first_error = store.get("a0/decision").decode().split("\n", 1)
    if decision != "fallback":
        if rank == 0:
            for l in held:
                print(l, flush=True)
            if decision == "accept" and any(rcs):
                print(f"[bench] the measurement was printed; exit codes {rcs} came from the ranks' teardown", file=sys.stderr, flush=True)
        if decision == "accept":
            return 0
        return rc if rc != 0 else max(rcs, key=abs)
    if rank == 0:
        for l in held:
            print("[bench attempt 0] " + l, file=sys.stderr, flush=True)
        print(f"[bench] first attempt failed ({first_error}); one fresh --conservative child per rank", file=sys.stderr, flush=True)
    rc, rcs, held = attempt(1, ["--conservative"], first_error)
    if rank == 0:
        for l in held:
            print(l, flush=True)
        if any(l.startswith('{"metric"') for l in held) and any(rcs):
            print(f"[bench] the fallback's measurement was printed; exit codes {rcs} came from the ranks' teardown", file=sys.stderr, flush=True)
            store.set("a1/accept", "1")
        else:
            store.set("a1/accept", "0")
    else:
        store.wait(["a1/accept"], timedelta(seconds=120))
    if store.get("a1/accept") == b"1":
        return 0
    return rc if rc != 0 else max(rcs, key=abs)


def attempt_store(attempt: int, world: int, timeout):
    """init_process_group arguments that give every ATTEMPT of a supervised run its own rendezvous namespace in the launcher's
    TCPStore (torch 2.10 hands all processes of a launch the same un-prefixed store: a second process group of the same ranks
    would read the first one's stale addresses and connect to the dead).  Without the launcher's store: env:// as it is."""
    if os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() != "true":
        return {}
    from torch.distributed import PrefixStore, TCPStore
    try:
        store = TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]), world, False, timeout)
    except Exception:  # noqa: BLE001 -- let init_process_group's own rendezvous report it
        return {}
    return dict(store=PrefixStore(f"/vorta_bench/attempt_{attempt}", store), rank=int(os.environ["RANK"]), world_size=world)


def stub_worker(args, rank: int, world: int, attempt: int) -> int:
    """CPU stand-in of a worker for the supervisor's rehearsal (VORTA_BENCH_STUB = comma-separated behaviours of the FIRST attempt:
    "fail" every rank exits 3 after a collective, "fail0" only rank 0 raises while the others wait in a collective, "hang1" rank 1
    never joins, "teardown1" rank 1 exits non-zero after the line was printed): a real gloo process group per attempt (its own rendezvous prefix), one all-reduce, ONE JSON line from rank 0."""
    import torch.distributed as dist
    from datetime import timedelta
    how = os.environ["VORTA_BENCH_STUB"].split(",") if attempt == 0 and not args.conservative else []
    if "hang1" in how and rank == 1:
        time.sleep(3600)
    pg_timeout = timedelta(seconds=float(os.environ.get("VORTA_BENCH_TIMEOUT_S", "120")))
    dist.init_process_group("gloo", timeout=pg_timeout, **attempt_store(attempt, world, pg_timeout))
    if "fail0" in how and rank == 0:
        print(json.dumps({"error": "stub: rank 0 failed before the collective", "n_gpus": world}), flush=True)
        os._exit(1)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    if "fail" in how:
        if rank == 0:
            print(json.dumps({"error": "stub: exchange self-check failed", "n_gpus": world}), flush=True)
        dist.destroy_process_group()
        return 3
    if rank == 0:
        print("[stub] a line that is not JSON", flush=True)
        print(json.dumps({"metric": "stub", "value": float(t.item()), "n_gpus": world, "conservative": bool(args.conservative),
                          **({"fallback": "conservative", "first_attempt_error": os.environ.get("VORTA_BENCH_FIRST_ATTEMPT_ERROR", "")}
                             if attempt > 0 else {})}), flush=True)
    dist.destroy_process_group()
    if "teardown1" in how and rank == 1:  # the measurement is out; this rank's exit fails
        return 7
    return 0


def main():
    # the host driver only supports dmabuf IPC: must be in the environment before HIP / HSA initialise
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="hunyuan-129f", choices=sorted(CONFIGS))
    ap.add_argument("--mix", default="uniform", choices=sorted(MIXES))
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp16", "fp8", "fp8pv", "i8pv", "auto8"],
                    help="fp8: bf16 inputs, converted to e4m3 inside the timed step (vorta_fp8_quantize_qkv), both "
                         "contractions on the fp8 MFMA, bf16 output; fp8pv: scores in bf16, P V in e4m3 (only v is converted, "
                         "inside the step); i8pv: scores on the int8 MFMA with one scale per row (k converted inside the step, q "
                         "by the attention kernel), P V in e4m3")
    ap.add_argument("--qkv-sets", type=int, default=2, help="distinct synthetic Q/K/V sets cycled over the layers")
    ap.add_argument("--sp-groups", default=os.environ.get("VORTA_SP_GROUPS", "1"),
                    help="N>1: exchange the local heads in this many slot groups so that the exchange of one group "
                         "overlaps the attention of another (1 = exchange, then attend: the default until a node run has "
                         "measured the links); auto = vorta_amd.ulysses.state.default_sp_groups: 1 at 3 heads per rank, 2-3 at "
                         "5, ranked on an EMULATED wire (profiles/r05_sp_groups_emulated.txt)")
    ap.add_argument("--emulate-rank", type=int, default=0, metavar="P",
                    help="on ONE GPU: the compute side of a P-way Ulysses step -- every layer as the rank that carries the "
                         "largest expert cost in THAT layer (a P-GPU step waits for its slowest rank layer by layer): its heads "
                         "over the whole sequence through the zero-copy receive layout, the send-side staging passes and the "
                         "un-permute of the output included, the transfers themselves left out.  `value` is an upper bound of "
                         "the P-GPU throughput; not a BASELINE line")
    ap.add_argument("--placement", default=os.environ.get("VORTA_SP_PLACEMENT", "auto"), choices=["auto", "even", "uneven", "split"],
                    help="N>1: heads per rank -- even: H/N on every rank (whole-head LPT under that constraint); uneven: the "
                         "ranks' head counts follow the layer's routes (LPT on the expert costs alone); auto (default): even "
                         "when N divides the heads (one receive layout for every layer; identical to uneven on the uniform "
                         "mix), uneven otherwise; split: uneven, then full-attention heads give a range of their QUERIES to the "
                         "lightest ranks until the heaviest is within 1 %% of the mean")
    ap.add_argument("--kv-splits", default="1",
                    help="N>1: 1 (default), a number, or 'auto': cut the keys of the full-attention / coreset launches of a rank "
                         "whose heads leave the chip under one round of workgroups (small models on many ranks); changes the "
                         "summation order, so never on by default")
    ap.add_argument("--conservative", action="store_true",
                    help="N>1 fallback: --placement auto --sp-groups 1 (one all_to_all_single per tensor for the whole layer), v "
                         "exchanged in 16 bits -- the oldest, most exercised form of the exchange; what a failed first attempt "
                         "falls back to by itself")
    ap.add_argument("--no-fallback", action="store_true",
                    help="N>1 under a launcher: a failed first attempt is final (default: every rank starts one fresh "
                         "--conservative child and its line is labelled \"fallback\": \"conservative\")")
    ap.add_argument("--no-selfcheck", action="store_true",
                    help="N>1: skip the exchange self-check that runs before the warm-up (integer-tagged q,k,v through layer "
                         "0's exchange with identity attention, compared exactly on every rank)")
    ap.add_argument("--no-v-wire", action="store_true",
                    help="N>1 with --dtype fp8: exchange v in 16 bits and convert it on the receive side (A/B; default: v "
                         "is converted on the send side and crosses the links as e4m3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gemm-ceiling", action="store_true",
                    help="skip the torch.matmul (hipBLASLt) context measurement reported beside roofline.frac")
    ap.add_argument("--sliding-block-rows", type=int, default=0, choices=[0, 128, 256],
                    help="query rows per workgroup of the sliding-tile launch (0 = library heuristic; 256 lets it join "
                         "the fused grid)")
    ap.add_argument("--level", default="attention", choices=["attention", "processor"],
                    help="attention (default, the BASELINE line): the routed-attention op on resident post-RoPE q,k,v with the "
                         "routes given (SURVEY.md section 8d).  processor: the call path the inference scripts take -- per "
                         "denoising step ONE vorta_route_plan call, then every block's attention processor __call__ "
                         "(q/k/v projections at the model's width, qk-norm + RoPE, device-resident routes from the plan, "
                         "routed attention, output projection) -- under torch.cuda.set_sync_debug_mode('error')")
    ap.add_argument("--experts", default="fused", choices=["fused", "serial", "concurrent"],
                    help="fused: the experts of a layer as ONE grid (vorta_attn_fwd_batch); serial: one launch per "
                         "expert on one stream; concurrent: experts on side HIP streams")
    args = ap.parse_args()

    cfg = dict(CONFIGS[args.config])
    if args.dtype:
        cfg["dtype"] = args.dtype
    if cfg.get("native_only"):  # --native_attention: the dense expert for every head
        args.mix = "all-full"
    if args.conservative:
        args.placement, args.sp_groups, args.no_v_wire = "auto", 1, True  # (auto = even wherever even exists)
    from vorta_amd.ulysses.state import default_sp_groups, resolve_placement  # the processors' rules (vorta_amd/attention/_sp.py)
    args.placement = resolve_placement(args.placement, cfg["heads"], max(args.emulate_rank or args.gpus, 1))
    if args.sp_groups == "auto":
        prec = {"fp8": True, "fp8pv": "fp8pv", "i8pv": "i8pv", "auto8": "auto8"}.get(cfg["dtype"], False)
        args.sp_groups = default_sp_groups(cfg["heads"] // max(args.emulate_rank or args.gpus, 1), prec)
    args.sp_groups = max(1, int(args.sp_groups))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.emulate_rank:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, as a CHILD process (never exec: nothing in this
        # process has touched the GPU yet, and nothing will), relay its output and exit with its code
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # VORTA_BENCH_FORCE_SP=1 (tests/test_hip_bench.py, with VORTA_SP_FORCE_COLLECTIVES=1): ONE rank takes the whole N > 1 path --
    # supervisor, process group on "nccl", the exchange's collectives on a world of one, self-check, breakdown -- the only
    # end-to-end rehearsal of `bench.py --gpus N` on RCCL that a one-GPU box allows (RCCL refuses two ranks on one device)
    sp_run = world > 1 or os.environ.get("VORTA_BENCH_FORCE_SP") == "1"
    if (sp_run and os.environ.get("VORTA_BENCH_WORKER") != "1"
            and os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() == "true"):
        # started by torch.distributed.run (the driver's N > 1 command, or `self_launch` above): this process stays GPU-free
        # and supervises a child; without the launcher's store (another launcher) the process is the worker itself
        rc = supervise(args)  # (None: the launcher's store could not be reached -- carry on as the worker)
        if rc is not None:
            sys.exit(rc)
    attempt = int(os.environ.get("VORTA_BENCH_ATTEMPT", "0"))
    if os.environ.get("VORTA_BENCH_STUB"):  # tests/test_bench_host.py: the supervisor's protocol on CPU ranks, no GPU anywhere
        sys.exit(stub_worker(args, rank, world, attempt))
    # VORTA_BENCH_BACKEND=gloo: 1-GPU rehearsal of the N>1 code path (all ranks share cuda:0, host-staged
    # messages); the driver's multi-GPU runs use RCCL ("nccl"), one rank per GPU
    backend = os.environ.get("VORTA_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    if sp_run:
        # a collective that never completes must not eat the caller's whole time limit and leave no line: the process group's
        # watchdog aborts the ranks after this long (VORTA_BENCH_TIMEOUT_S), and `guarded` below turns that into one JSON line
        from datetime import timedelta
        pg_timeout = timedelta(seconds=float(os.environ.get("VORTA_BENCH_TIMEOUT_S", "120")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=pg_timeout, **attempt_store(attempt, world, pg_timeout))
        else:
            dist.init_process_group(backend, timeout=pg_timeout, **attempt_store(attempt, world, pg_timeout))

    # what the process group saw: one entry per rank (proof that N ranks on N devices ran the step over RCCL)
    props = torch.cuda.get_device_properties(dev_index)
    me = {"rank": rank, "local_rank": local_rank, "device": dev_index, "name": props.name,
          "uuid": str(getattr(props, "uuid", "")), "host": os.uname().nodename, "pid": os.getpid()}
    ranks_seen = [me]
    if sp_run:
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, me)

    from vorta_amd import ops
    from vorta_amd.routed import HeadRouting, RoutedGeometry, routed_attention

    dt = torch.float16 if cfg["dtype"] == "fp16" else torch.bfloat16  # (the 8-bit paths take and return bf16)
    fp8 = True if cfg["dtype"] == "fp8" else (cfg["dtype"] if cfg["dtype"] in ("fp8pv", "i8pv", "auto8") else False)
    # mixed precision: half of a layer's FLOPs run at the 16-bit rate, half at the e4m3 rate: harmonic mean of the peaks
    # (int8 scores run at the e4m3 MFMA rate: the all-8-bit peak)
    peak = PEAK_MFMA_FP8_TFLOPS if fp8 in (True, "i8pv", "auto8") else (2.0 / (1.0 / PEAK_MFMA_TFLOPS + 1.0 / PEAK_MFMA_FP8_TFLOPS)
                                                               if fp8 == "fp8pv" else PEAK_MFMA_TFLOPS)
    H, L, T, te = cfg["heads"], cfg["layers"], cfg["text"], cfg["text_valid"]
    S = cfg["latent"][0] * cfg["latent"][1] * cfg["latent"][2]
    hy = cfg["model"] == "hunyuan"
    P = world
    concurrent, fused = args.experts == "concurrent", args.experts == "fused"

    # ---- per-layer routing (and, for P>1, the head -> rank placement that balances expert cost) ----
    layer_ids = [layer_experts(cfg, args.mix, l) for l in range(L)]
    step_flops = 0.0
    for e in layer_ids:
        f, per_head = algorithmic_flops(cfg, e)
        step_flops += f * cfg["fwd_per_step"]

    emu = args.emulate_rank
    if emu:
        if world != 1:
            raise SystemExit("--emulate-rank runs on one GPU")
        P = emu
    proc_info = None
    if args.level == "processor":
        if P != 1:
            raise SystemExit("--level processor runs on one GPU")
        one_step, fp_tensor, layer_ids, proc_info = processor_level(cfg, args.mix, dev, dt, fp8)
    elif P == 1 and not sp_run:
        geom = RoutedGeometry(cfg["latent"], cfg["tile"], cfg["window"], cfg["group"], cfg["rate"], dev)
        routings = [HeadRouting.from_expert_ids(e, dev) for e in layer_ids]
        sets = []
        for i in range(args.qkv_sets):
            gen = torch.Generator(device=dev).manual_seed(1234 + i)
            sets.append(tuple(torch.randn((1, H, S + T, 128), generator=gen, device=dev, dtype=dt) for _ in range(3)))
        out = torch.empty_like(sets[0][0])
        # e4m3 operand buffers reused by every layer (the conversion itself runs per layer, inside the step)
        f8buf = (ops.fp8_quantize_qkv(*(x[0] for x in sets[0])) if fp8 is True else
                 ops.fp8_quantize_v(sets[0][2][0]) if fp8 == "fp8pv" else
                 (ops.fp8_quantize_v(sets[0][2][0]), ops.i8_quantize_k(sets[0][0][0], sets[0][1][0])) if fp8 in ("i8pv", "auto8") else None)
        if te:
            geom.sta_tables(te)  # built once per prompt, outside the step (pipeline_hunyuan.py:378-392)

        def one_step():
            for _ in range(cfg["fwd_per_step"]):
                for l in range(L):
                    q, k, v = sets[l % len(sets)]
                    routed_attention(q, k, v, routings[l], geom, model=cfg["model"], text_len=T, text_valid=te, out=out,
                                     concurrent=concurrent, fused=fused, sliding_block_rows=args.sliding_block_rows,
                                     fp8=fp8, fp8_operands=f8buf)
    else:
        from vorta_amd import ulysses
        sp = ulysses.UlyssesRoutedAttention(cfg, layer_ids, per_head, dev, dt, rank, P,
                                            concurrent=concurrent, fused=fused, sliding_block_rows=args.sliding_block_rows,
                                            groups=args.sp_groups, loopback=bool(emu), fp8=fp8, v_wire=not args.no_v_wire,
                                            placement=args.placement, heaviest_rank=bool(emu),
                                            kv_splits=args.kv_splits if args.kv_splits == "auto" else int(args.kv_splits))

        def one_step():
            for _ in range(cfg["fwd_per_step"]):
                for l in range(L):
                    sp.layer(l)

    def barrier():
        if sp_run:
            dist.barrier()
        torch.cuda.synchronize()

    # N > 1: before anything is timed, prove that the exchange moves the right bytes on THIS transport (RCCL on the driver's
    # node, gloo in the one-GPU rehearsals): vorta_amd/ulysses/engine.py exchange_selfcheck on layer 0's placement
    selfcheck = None
    if sp_run and not emu and not args.no_selfcheck:
        # (VORTA_SP_SELFCHECK_BREAK=1, tests only: breaks the FIRST attempt, so the fallback has something to recover from)
        selfcheck = sp.selfcheck(0, break_order=os.environ.get("VORTA_SP_SELFCHECK_BREAK") == "1" and attempt == 0)
        if not selfcheck["ok"]:
            if rank == 0:
                print(json.dumps({"error": "exchange self-check failed: the head exchange did not deliver the expected rows",
                                  "exchange_selfcheck": selfcheck, "n_gpus": world, "backend": backend}), flush=True)
            barrier()
            dist.destroy_process_group()
            sys.exit(3)

    for _ in range(args.warmup):
        one_step()
    barrier()
    tl = ops.Timeline()
    ops.set_timeline(tl)
    if proc_info is not None:  # any host-device synchronisation inside the step raises
        torch.cuda.set_sync_debug_mode("error")
    t0 = time.perf_counter()
    try:
        for _ in range(args.steps):
            one_step()
    finally:
        if proc_info is not None:
            torch.cuda.set_sync_debug_mode("default")
    barrier()
    elapsed = time.perf_counter() - t0
    ops.set_timeline(None)
    if sp_run:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed * 1e3 / args.steps
    # order-independent fingerprint of the last layer's output (all ranks): lets two runs of one workload -- other slot
    # groups, v on the wire or not -- be compared bit for bit from their JSON lines
    fp_t = (fp_tensor() if proc_info is not None else out if (P == 1 and not sp_run) else sp.out_shard).contiguous().view(torch.int16).to(torch.int64)
    fingerprint = (fp_t * (torch.arange(fp_t.numel(), device=dev).view(fp_t.shape) % 8191 + 1)).sum().reshape(1)
    if sp_run:
        fingerprint = fingerprint * (rank + 1)
        dist.all_reduce(fingerprint)
    fingerprint = int(fingerprint.item())
    tokens = S * cfg["fwd_per_step"]

    # ---- N > 1: what the exchange costs and how much of it the step hides (after the timed region, never part of `value`) ----
    exchange = None
    if sp_run and not emu:
        exchange = exchange_breakdown(sp, cfg, args, ms_per_step, barrier, dist, dev, world, fp8, backend)

    # ---- roofline of the dominant kernel symbol (largest share of the timed region) ----
    summ = tl.summary()  # keyed by (expert tag, kernel symbol)
    by_kernel = {}
    for (tag, sym), v in summ.items():
        d = by_kernel.setdefault(sym, dict(ms=0.0, flops=0.0, launches=0))
        d["ms"] += v["ms"]; d["flops"] += v["flops"]; d["launches"] += v["launches"]
    dom_sym = max(by_kernel, key=lambda k: by_kernel[k]["ms"])
    dom = by_kernel[dom_sym]
    if proc_info is not None and dom["flops"] == 0.0:
        # device-resident routes: the launches do not know their head counts on the host; every attention launch of a
        # layer is in the fused grid, whose work is the layer's algorithmic work
        dom["flops"] = step_flops * args.steps
    achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
    roofline = {"bound": "mfma", "kernel": dom_sym,
                "achieved": round(achieved, 1), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": None,
                "launches": dom["launches"], "avg_launch_ms": round(dom["ms"] / max(dom["launches"], 1), 4),
                "flops_per_launch": dom["flops"] / max(dom["launches"], 1),
                "share_of_step": round(dom["ms"] / (ms_per_step * args.steps), 3)}
    # bytes per launch from the PMC passes committed under profiles/ (bench.py cannot run rocprofv3 on itself), keyed by
    # workload: `traffic` = bytes leaving the L2s (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md 'HBM'); the Infinity
    # Cache sits behind that interface.  The NEWEST table that holds this workload is used (r05b > r05 > r04b > ...) and named,
    # with the git head of the tree it was taken on when the table records it: the line says which binary its traffic describes.
    # (A DRAM-side estimate from the memory controllers' activity is quoted only from a pass of the SAME round as the table.)
    try:
        import glob
        import re
        tables = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*_pmc_traffic.json")),
                        key=lambda f: re.match(r"r(\d+)([a-z]*)_", os.path.basename(f)).groups(), reverse=True)
        wl, pmc_file, table = None, None, None
        for f in tables:
            table = json.load(open(f))
            wl = table["workloads"].get(f"{args.config} {args.mix} {cfg['dtype']}")
            if wl is not None:
                pmc_file = os.path.basename(f)
                break
        if wl is not None and world == 1 and not sp_run and not emu and proc_info is None:
            base = dom_sym.split("<")[0]
            hit = [v for k_, v in wl["kernels"].items() if base in k_]
            if len(hit) == 1:
                roofline["traffic"] = hit[0]["l2_miss_bytes_per_launch"]
                roofline["traffic_unit"] = ("bytes per launch leaving the L2s (FETCH_SIZE x2 + WRITE_SIZE, "
                                            f"profiles/{pmc_file})")
                roofline["traffic_source"] = f"profiles/{pmc_file}"
                roofline["traffic_source_head"] = table.get("head")  # git head of the profiled tree (None: older tables)
                roofline["traffic_over_minimum"] = round(hit[0]["l2_miss_bytes_per_launch"] /
                                                         wl["algorithmic_min_bytes_per_fused_launch"], 2)
                rnd = re.match(r"(r\d+)", pmc_file).group(1)
                umc = sorted(glob.glob(os.path.join(ROOT, "profiles", f"{rnd}*_umc_activity_{args.config}_{cfg['dtype']}.json")))
                if umc and args.mix == "uniform":
                    u = json.load(open(umc[-1]))
                    roofline["traffic_dram_estimate"] = u["from_percent"]["bytes_per_layer"]
                    roofline["traffic_dram_unit"] = ("HBM bytes per launch from the memory controllers' activity (coarse: "
                                                     f"integer percent; tools/umc_activity.py, profiles/{os.path.basename(umc[-1])})")
    except Exception:
        pass
    per_tag = {f"{tag}: {sym}": {"launches": v["launches"], "avg_ms": round(v["ms"] / v["launches"], 4),
                                 "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["ms"] > 0 else 0.0}
               for (tag, sym), v in sorted(summ.items())}

    res = {
        "metric": "video_tokens_per_sec (routed-attention denoising step, HunyuanVideo 720p 129f)"
        if args.config == "hunyuan-129f" else f"video_tokens_per_sec (routed-attention denoising step, {args.config})",
        "value": round(tokens / (ms_per_step * 1e-3), 1), "unit": "video_tokens/s",
        **({"emulated_rank_of": emu, "note": "the heaviest rank's compute (layer by layer) of a %d-GPU Ulysses step on one GPU (no transfers): an "
            "upper bound of the %d-GPU value, not a measurement of it" % (emu, emu)} if emu else {}),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 2),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": cfg["dtype"], "data": "synthetic",
        "backend": (backend if sp_run else None), "output_fingerprint": fingerprint,
        **({"exchange_selfcheck": selfcheck} if selfcheck is not None else {}),
        **({"fallback": "conservative", "first_attempt_error": os.environ.get("VORTA_BENCH_FIRST_ATTEMPT_ERROR", "")}
           if attempt > 0 else {}),
        **({"exchange": exchange} if exchange is not None else {}),
        "process_group": {"world_size": dist.get_world_size() if sp_run else 1,
                          "backend": dist.get_backend() if sp_run else None,
                          "distinct_devices": len({(r["host"], r["uuid"] or r["device"]) for r in ranks_seen}),
                          "ranks": ranks_seen},
        "config": {"workload": f"{args.config}: {cfg['model']} latent {cfg['latent']} S={S} text {T}/{te} H={H} "
                               f"layers={L} x{cfg['fwd_per_step']} fwd/step; tile {cfg['tile']} window {cfg['window']} "
                               f"coreset {cfg['group']} r={cfg['rate']}; routing mix '{args.mix}' (rng 1234+layer)",
                   "parallelism": "single GPU" if (P == 1 and not sp_run) else (f"heaviest rank (layer by layer) of ulysses sp{P}, emulated on one GPU, no transfers"
                                                               if emu else f"ulysses sp{P} (RCCL all-to-all over xGMI)")
                   + (f", {args.sp_groups} overlapped slot groups" if args.sp_groups > 1 else "")
                   + (f", key splits {args.kv_splits} (per layer: {sorted(set(sp.kv_splits))})" if P > 1 and args.kv_splits != "1" else "")
                   + (f", {args.placement} head placement (heaviest rank / mean cost, worst layer: "
                      f"{max(sp.max_over_mean):.3f}"
                      + (f"; heads split by query range: {sum(len(x) - H for x in sp.orders)} extra parts over {L} layers"
                         if args.placement == "split" else "") + ")" if P > 1 else ""),
                   "experts": {"fused": "one fused grid per layer", "serial": "one launch per expert",
                               "concurrent": "experts on side streams"}[args.experts],
                   **({"fp8": "e4m3 q,k,v and probabilities on the fp8 MFMA; conversion (per-head scales, key centring "
                              + ("on" if __import__("vorta_amd.routed", fromlist=["x"]).FP8_CENTER_K else "off")
                              + ") inside the timed step; bf16 in / out"} if fp8 is True else
                      {"fp8pv": "scores on the bf16 MFMA, probabilities and v in e4m3 on the fp8 MFMA; v converted inside the "
                                "timed step; bf16 in / out; roofline.peak = harmonic mean of the two MFMA peaks (3 333)"}
                      if fp8 == "fp8pv" else
                      {"i8pv": "scores on the int8 MFMA (k: centred, channel-balanced, one scale per row, converted inside the "
                               "timed step; q: converted by the attention kernel), probabilities and v in e4m3 on the fp8 MFMA; "
                               "bf16 in / out"} if fp8 == "i8pv" else {}),
                   **({"call_path": proc_info, "ms_per_layer": round(ms_per_step / (L * cfg["fwd_per_step"]), 3)}
                      if proc_info is not None else {}),
                   "step_algorithmic_pflop": round(step_flops / 1e15, 3),
                   "step_tflops_per_gpu": round(step_flops / (ms_per_step * 1e-3) / 1e12 / (emu or world), 1)},
        "roofline": roofline,
        "per_launch": per_tag,
        # behaviour switches in effect: every VORTA_* variable of the environment and the library's build info (a
        # variant build lists its -D flags there)
        "switches": {"env": {k_: v_ for k_, v_ in sorted(os.environ.items()) if k_.startswith("VORTA_")},
                     "library": __import__("vorta_amd._C", fromlist=["lib"]).lib().vorta_build_info().decode()},
    }
    if world == 1 and not sp_run and not args.no_gemm_ceiling:
        # context for `frac`: what a plain library GEMM (hipBLASLt via torch.matmul, same dtype) sustains on THIS
        # box right now -- the practical MFMA ceiling under the chip's power/clock management (measured after the
        # timed region, never part of `value`)
        try:
            n = 16384
            a = torch.randn((n, n), device=dev).to(dt)
            b = torch.randn((n, n), device=dev).to(dt)
            mm = lambda: a @ b
            if fp8 is True:  # both contractions in e4m3: the library's e4m3 GEMM (torch._scaled_mm, unit scales)
                a, b = a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn).t().contiguous().t()
                one = torch.ones((), device=dev)
                mm = lambda: torch._scaled_mm(a, b, scale_a=one, scale_b=one, out_dtype=torch.bfloat16)
                roofline["library_gemm_dtype"] = "e4m3"
            for _ in range(2):
                mm()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                mm()
            e1.record()
            torch.cuda.synchronize()
            gemm = 2.0 * n ** 3 * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e12
            roofline["library_gemm_tflops"] = round(gemm, 1)
            roofline["frac_of_library_gemm"] = round(achieved / gemm, 4) if gemm > 0 else None
            del a, b
        except Exception as exc:  # context only: never let it cost the bench line
            roofline["library_gemm_tflops"] = None
            roofline["library_gemm_error"] = f"{type(exc).__name__}: {exc}"[:200]
    if world == 1 and not sp_run and not args.no_gemm_ceiling and not emu and proc_info is None:
        borrowed_dense(cfg, S, te, H, L, dev, roofline, res, achieved, layer_ids)
    if rank == 0:
        if world == 1 and not sp_run and not args.no_cpu_baseline and not emu and proc_info is None:
            res["cpu_baseline"] = cpu_baseline(cfg, layer_ids, cfg["dtype"])
        abandoned = res.pop("_abandoned_thread", False)
        print(json.dumps(res), flush=True)
        if abandoned:  # the line is out; do not wait for a thread that may never return
            os._exit(0)
    if sp_run:
        dist.destroy_process_group()


def borrowed_dense(cfg, S, te, H, L, dev, roofline, res, achieved, layer_ids=None, budget_s: float = 45.0):
    """Context beside `library_gemm_tflops`, measured after the timed region and never part of `value`: the kernel the
    REFERENCE borrows for its dense expert on this box -- torch.nn.functional.scaled_dot_product_attention, the call at
    /root/reference/vorta/attention/hunyuan.py:169-176 (wan.py:142-145) -- on a sample of 8 heads x (S + T_eff)^2 in 16 bits,
    and this build's own dense launch (ops.attn_fwd) on the same tensors.  torch-ROCm is not the reference (nothing of the
    reference travels); it is the attention kernel the reference's code would have called here.  `step_ms_if_borrowed_dense` =
    the all-full (--native_attention) step through that kernel, EXTRAPOLATED from the sample: heads x layers x forwards."""
    import torch.nn.functional as F
    from vorta_amd import ops
    t_start = time.perf_counter()
    try:
        dt16 = torch.float16 if cfg["dtype"] == "fp16" else torch.bfloat16
        N, hs = S + te, 8
        gen = torch.Generator(device=dev).manual_seed(99)
        q, k, v = (torch.randn((1, hs, N, 128), generator=gen, device=dev, dtype=dt16) for _ in range(3))
        flops = 4.0 * N * N * 128

        def timed(fn, heads, reps):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps / heads  # ms per head

        # one head first: if the library kernel is very slow here the 8-head sample would eat the budget
        t0 = time.perf_counter()
        ms1 = timed(lambda: F.scaled_dot_product_attention(q[:, :1], k[:, :1], v[:, :1]), 1, 1)
        heads = hs if (time.perf_counter() - t0) * hs * 1.5 < budget_s - (time.perf_counter() - t_start) else 1
        ms_sdpa = timed(lambda: F.scaled_dot_product_attention(q[:, :heads], k[:, :heads], v[:, :heads]), heads, 2) if heads > 1 else ms1
        o = torch.empty((1, hs, N, 128), device=dev, dtype=dt16)
        ms_own = timed(lambda: ops.attn_fwd(q[0, :heads], k[0, :heads], v[0, :heads], o[0, :heads], n_q=N, n_kv=N), heads, 2)
        sdpa_tf, own_tf = flops / (ms_sdpa * 1e-3) / 1e12, flops / (ms_own * 1e-3) / 1e12
        roofline["library_sdpa_tflops"] = round(sdpa_tf, 1)
        roofline["library_sdpa_sample"] = (f"F.scaled_dot_product_attention, {heads} heads x ({N})^2 x 128, {cfg['dtype'] if cfg['dtype'] in ('fp16', 'bf16') else 'bf16'}, "
                                           f"torch {torch.__version__}")
        roofline["frac_of_library_sdpa"] = round(achieved / sdpa_tf, 4) if sdpa_tf > 0 else None
        roofline["own_dense_tflops_same_sample"] = round(own_tf, 1)
        roofline["own_dense_over_library_sdpa"] = round(own_tf / sdpa_tf, 3) if sdpa_tf > 0 else None
        fwd = cfg["fwd_per_step"]
        res["config"]["step_ms_if_borrowed_dense"] = round(ms_sdpa * H * L * fwd, 1)
        res["config"]["step_ms_own_dense"] = round(ms_own * H * L * fwd, 1)
        res["config"]["borrowed_dense_note"] = ("all-full mix (--native_attention) extrapolated from the sample: ms per head x "
                                                f"{H} heads x {L} layers x {fwd} forwards; SDPA = the kernel the reference calls "
                                                "(hunyuan.py:169-176), torch-ROCm build on this box")
        if layer_ids is not None and time.perf_counter() - t_start < budget_s:
            # in a daemon thread with a deadline: torch.compile (its worker processes, Triton) is the one thing on this path that
            # could HANG rather than fail, and a context measurement must not be able to cost the bench line
            import threading
            left = budget_s - (time.perf_counter() - t_start)
            # the thread writes into a dict of its OWN: an abandoned thread that finishes late must not touch `res`
            # while the main thread serialises it
            private = {"config": {}, "ms_per_step": res["ms_per_step"]}
            th = threading.Thread(target=borrowed_routed, daemon=True,
                                  args=(cfg, S, te, L, layer_ids, q, k, v, ms_sdpa, timed, private, left, dev))
            th.start()
            th.join(max(left, 1.0) + 60.0)
            if th.is_alive():
                res["config"]["step_ms_if_borrowed_routed"] = None
                res["config"]["borrowed_routed_error"] = "no answer within the deadline (torch.compile of flex_attention); abandoned"
                res["_abandoned_thread"] = True
            else:
                res["config"].update(private["config"])
    except Exception as exc:  # context only: never let it cost the bench line
        roofline["library_sdpa_tflops"] = None
        roofline["library_sdpa_error"] = f"{type(exc).__name__}: {exc}"[:200]


def window_tile_matrix(latent, tile, window, dev):
    """(n_tiles, n_tiles) bool: may a query tile see a key tile?  Tiles in raster order over the tile grid; per dimension the
    window centre is the query tile's coordinate clamped so that the window stays inside the grid (SURVEY.md section 8 A8;
    tests/test_bench_host.py compares it with the oracle's table).  Used only by the borrowed-kernel context measurement."""
    n = [l // t for l, t in zip(latent, tile)]
    idx = torch.arange(n[0] * n[1] * n[2], device=dev)
    co = [idx // (n[1] * n[2]), (idx // n[2]) % n[1], idx % n[2]]
    ok = torch.ones((idx.numel(), idx.numel()), dtype=torch.bool, device=dev)
    for d in range(3):
        half = window[d] // 2
        c = co[d].clamp(min=half, max=max(n[d] - 1 - half, half))
        ok &= (co[d][None, :] - c[:, None]).abs() <= half
    return ok


def borrowed_routed(cfg, S, te, L, layer_ids, q, k, v, ms_full, timed, res, budget_s, dev=None):
    """The other two library kernels the reference's routed path borrows, on the sample tensors of `borrowed_dense`: SDPA on the
    coreset expert's pooled sequence (hunyuan.py:441-448) and torch's COMPILED flex_attention under the sliding-tile block mask
    (sliding_attn_flex.py:137-211; mask restated from SURVEY.md section 8 A8: tile-major order, clamped window, text rules).
    `step_ms_if_borrowed_routed` = this run's head -> expert assignment through those three kernels, per-head times x head
    counts; the reference's torch-side pooling / tiling / combine passes are NOT included: a lower bound of its step on this
    box.  Context only (torch-ROCm is not the reference); tools/bench_borrowed_routed.py is the stand-alone form."""
    try:
        import torch.nn.functional as F
        from torch.nn.attention.flex_attention import create_block_mask, flex_attention
        if dev is not None:
            torch.cuda.set_device(dev)  # (a thread of its own: the current device is per thread)
        dev, hs = q.device, q.shape[1]
        g = cfg["group"][0] * cfg["group"][1] * cfg["group"][2]
        n_low = (S // g) * (1 + int(g * (1 - cfg["rate"])) - 1) + te
        ms_low = timed(lambda: F.scaled_dot_product_attention(q[:, :, :n_low], k[:, :, :n_low], v[:, :, :n_low]), hs, 2)
        tok = cfg["tile"][0] * cfg["tile"][1] * cfg["tile"][2]
        tiles = window_tile_matrix(cfg["latent"], cfg["tile"], cfg["window"], dev)
        T = cfg["text"]

        def mask_mod(b, h, qi, ki):
            vq, vk = qi < S, ki < S
            video = vq & vk & tiles[torch.where(vq, qi // tok, 0), torch.where(vk, ki // tok, 0)]
            return video | (vq & (ki >= S) & (ki < S + te)) | ((qi >= S) & (qi < S + te) & (ki < S + te))

        N = S + T
        if q.shape[2] < N:  # (the dense sample holds S + T_eff rows)
            pad = lambda x: torch.cat([x, torch.zeros((1, hs, N - x.shape[2], 128), device=dev, dtype=x.dtype)], 2)
            q, k, v = pad(q), pad(k), pad(v)
        t0 = time.perf_counter()
        bm = torch.compile(create_block_mask)(mask_mod, None, None, N, N, device=dev)
        flex = torch.compile(flex_attention, dynamic=False)
        flex(q, k, v, block_mask=bm)
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        if first > budget_s:
            raise TimeoutError(f"first compiled flex_attention call took {first:.0f} s")
        ms_sl = timed(lambda: flex(q, k, v, block_mask=bm), hs, 2)
        cnt = [sum(int((ids == e).sum()) for ids in layer_ids) for e in range(3)]
        fwd = cfg["fwd_per_step"]
        c = res["config"]
        c["step_ms_if_borrowed_routed"] = round((cnt[0] * ms_full + cnt[1] * ms_low + cnt[2] * ms_sl) * fwd, 1)
        c["borrowed_routed_over_own_step"] = round(c["step_ms_if_borrowed_routed"] / res["ms_per_step"], 3)
        c["borrowed_routed_ms_per_head"] = {"full_sdpa": round(ms_full, 3), "coreset_sdpa": round(ms_low, 3),
                                            "sliding_flex": round(ms_sl, 3), "flex_first_call_s": round(first, 1)}
        c["borrowed_routed_note"] = ("this run's routes through the library kernels the reference calls (SDPA full + SDPA on the pooled "
                                     "sequence + compiled flex_attention under the tile block mask), per-head times x head counts over "
                                     f"{L} layers x {fwd} forwards; its torch-side pooling / tiling / combine passes left out: a lower bound")
    except Exception as exc:  # context only
        res["config"]["step_ms_if_borrowed_routed"] = None
        res["config"]["borrowed_routed_error"] = f"{type(exc).__name__}: {exc}"[:200]


def guarded():
    """main() with one promise: a failure leaves ONE JSON line naming it on rank 0's stdout and a non-zero exit code (a failed
    GPU process exits; it never re-executes itself)."""
    try:
        main()
    except SystemExit:
        raise
    except BaseException as exc:  # noqa: BLE001
        import traceback
        traceback.print_exc()
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps({"error": f"{type(exc).__name__}: {exc}"[:600], "n_gpus": int(os.environ.get("WORLD_SIZE", "1")),
                              "argv": sys.argv[1:]}), flush=True)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(1)  # not sys.exit: a rank stuck in a collective's destructor would hang the exit


if __name__ == "__main__":
    guarded()
