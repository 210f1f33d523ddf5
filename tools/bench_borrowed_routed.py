# What would the REFERENCE's routed attention step cost on this box?  Nothing of the reference travels, but the kernels its three
# experts BORROW are in torch-ROCm: F.scaled_dot_product_attention for the dense expert (vorta/attention/hunyuan.py:169-176) and
# for the coreset expert's pooled sequence (hunyuan.py:441-448), torch's compiled flex_attention under a block mask for the
# sliding-tile expert (vorta/attention/sliding_attn_flex.py:137-211).  This script times those three library calls on the headline
# shapes (synthetic N(0,1) tensors, 8 heads each) and extrapolates one denoising step of attention: (full + coreset + sliding) x
# layers x forwards.  It leaves OUT what the reference does around them in torch -- similarity ranking, gather / scatter of the
# pooled rows, tile / untile permutations, the index-put that combines the experts -- so the figure is a LOWER bound of the
# borrowed path's step.  The mask is restated from SURVEY.md §8 A8 (tile-major order, clamped window: bench.window_tile_matrix, checked
# against the oracle's table on the CPU; text rules below) and one flex output row is checked against a direct softmax.
# A measurement of library kernels for context: the stand-alone form of bench.py's `step_ms_if_borrowed_routed`.
# usage: python tools/bench_borrowed_routed.py [--config hunyuan-129f] [--heads 8] > gpurun_out/borrowed_routed.json
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import window_tile_matrix as window_tiles  # noqa: E402  (the one restatement of the tile window rule)

CONFIGS = {  # latent, tile, window, coreset keep ratio, text pad / valid, heads per expert (uniform mix), layers, forwards, dtype
    "hunyuan-129f": dict(latent=(33, 45, 80), tile=(11, 9, 8), window=(3, 3, 3), s_low=59400, T=256, te=96, heads=(8, 8, 8),
                         layers=60, fwd=1, dtype=torch.float16),
    "wan14b-81f": dict(latent=(21, 45, 80), tile=(7, 9, 8), window=(3, 3, 3), s_low=37800, T=0, te=0, heads=(14, 13, 13),
                       layers=40, fwd=2, dtype=torch.bfloat16),
}


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="hunyuan-129f", choices=sorted(CONFIGS))
    ap.add_argument("--heads", type=int, default=8, help="heads per timed call (per-head time x the mix's head counts)")
    a = ap.parse_args()
    c = CONFIGS[a.config]
    dev = torch.device("cuda:0")
    S = c["latent"][0] * c["latent"][1] * c["latent"][2]
    T, te, dt, hs = c["T"], c["te"], c["dtype"], a.heads
    tok = c["tile"][0] * c["tile"][1] * c["tile"][2]
    gen = torch.Generator(device=dev).manual_seed(7)
    q, k, v = (torch.randn((1, hs, S + T, 128), generator=gen, device=dev, dtype=dt) for _ in range(3))
    out = {"config": a.config, "dtype": str(dt), "heads_per_call": hs, "torch": torch.__version__, "device": torch.cuda.get_device_name(0)}

    # dense expert: SDPA on the valid rows
    N = S + te
    ms_full = timed(lambda: F.scaled_dot_product_attention(q[:, :, :N], k[:, :, :N], v[:, :, :N])) / hs
    out["full_ms_per_head"] = round(ms_full, 3)
    out["full_tflops"] = round(4.0 * N * N * 128 / (ms_full * 1e-3) / 1e12, 1)
    print(f"full: {ms_full:.2f} ms/head", file=sys.stderr, flush=True)

    # coreset expert: SDPA on the pooled sequence + text (the pooling passes themselves are left out)
    Nl = c["s_low"] + te
    ms_low = timed(lambda: F.scaled_dot_product_attention(q[:, :, :Nl], k[:, :, :Nl], v[:, :, :Nl])) / hs
    out["coreset_ms_per_head"] = round(ms_low, 3)
    out["coreset_tflops"] = round(4.0 * Nl * Nl * 128 / (ms_low * 1e-3) / 1e12, 1)
    print(f"coreset: {ms_low:.2f} ms/head", file=sys.stderr, flush=True)

    # sliding-tile expert: compiled flex_attention under the block mask (tile-major token order, text at the end)
    try:
        from torch.nn.attention.flex_attention import create_block_mask, flex_attention
        tiles = window_tiles(c["latent"], c["tile"], c["window"], dev)

        def mask_mod(b, h, qi, ki):
            vq, vk = qi < S, ki < S
            tq = torch.where(vq, qi // tok, 0)
            tk = torch.where(vk, ki // tok, 0)
            video = vq & vk & tiles[tq, tk]
            video_to_text = vq & (ki >= S) & (ki < S + te)
            text_to_all = (qi >= S) & (qi < S + te) & (ki < S + te)
            return video | video_to_text | text_to_all

        t0 = time.perf_counter()
        bm = torch.compile(create_block_mask)(mask_mod, None, None, S + T, S + T, device=dev)
        flex = torch.compile(flex_attention, dynamic=False)
        o = flex(q, k, v, block_mask=bm)
        torch.cuda.synchronize()
        out["flex_first_call_s"] = round(time.perf_counter() - t0, 1)
        print(f"flex compiled in {out['flex_first_call_s']} s", file=sys.stderr, flush=True)
        ms_sl = timed(lambda: flex(q, k, v, block_mask=bm)) / hs
        n_kv = int(tiles[tiles.shape[0] // 2].sum().item()) * tok
        fl = 4.0 * 128 * (S * (n_kv + te) + te * (S + te))  # interior tile: an upper bound of the mean
        out["sliding_ms_per_head"] = round(ms_sl, 3)
        out["sliding_tflops_interior_count"] = round(fl / (ms_sl * 1e-3) / 1e12, 1)
        out["block_mask_sparsity_pct"] = round(float(bm.sparsity()), 2)
        # sanity: one query row against a direct softmax over its allowed keys
        r = S // 2 + 5
        allow = mask_mod(0, 0, torch.tensor(r, device=dev), torch.arange(S + T, device=dev))
        s = (q[0, 0, r].float() @ k[0, 0].float().T) / 128 ** 0.5
        p = torch.softmax(torch.where(allow, s, float("-inf")), -1)
        ref = p @ v[0, 0].float()
        out["flex_row_check_max_err"] = float((o[0, 0, r].float() - ref).abs().max())
        print(f"sliding: {ms_sl:.2f} ms/head", file=sys.stderr, flush=True)
    except Exception as exc:  # noqa: BLE001 -- the compiled kernel may not build on this box: say so
        ms_sl = None
        out["sliding_error"] = f"{type(exc).__name__}: {exc}"[:400]

    h0, h1, h2 = c["heads"]
    if ms_sl is not None:
        layer = h0 * ms_full + h1 * ms_low + h2 * ms_sl
        out["borrowed_routed_layer_ms"] = round(layer, 2)
        out["borrowed_routed_step_ms"] = round(layer * c["layers"] * c["fwd"], 1)
    out["borrowed_dense_step_ms"] = round((h0 + h1 + h2) * ms_full * c["layers"] * c["fwd"], 1)
    out["note"] = ("library attention kernels only (SDPA, compiled flex_attention), per-head time x head counts x layers x forwards; "
                   "the reference's torch-side pooling / tiling / combine passes are not included: a lower bound of its step")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
