#!/usr/bin/env python
"""Path-level accuracy of the ROUTED operator against native (all-dense) attention on synthetic q, k, v WITH
spatio-temporal structure (tests/_structured_inputs.py), at the BASELINE geometries, per routing mix and noise share.

    python tools/structured_psnr.py [--config hunyuan-129f] [--heads 12] > profiles/r06_structured_psnr_<config>.txt

north_star asks PSNR >= 40 dB against --native_attention; on white noise the METHOD gives 22 dB (no expert has anything to
exploit).  A trained router cannot exist in the offline image, so each head gets the structure its expert assumes -- what the
router is trained to find (/root/reference/vorta/attention/hunyuan.py:562-605) -- and, as the control, the other expert's.
Both sides of every comparison run this build's HIP kernels in bf16 (dense_attention = the --native_attention kernel)."""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch


def main():
    import bench
    from _structured_inputs import structured_layer
    from vorta_amd.routed import HeadRouting, RoutedGeometry, dense_attention, routed_attention
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="hunyuan-129f", choices=sorted(bench.CONFIGS))
    ap.add_argument("--heads", type=int, default=12, help="heads of the sample layer (the mix's fractions of them per expert)")
    ap.add_argument("--precision", default="native", choices=["native", "fp8pv", "i8pv", "auto8"])
    args = ap.parse_args()
    cfg = bench.CONFIGS[args.config]
    dev = torch.device("cuda", 0)
    latent, tile, window, group = cfg["latent"], cfg["tile"], cfg["window"], cfg["group"]
    H = args.heads
    geom = RoutedGeometry(latent, tile, window, group, cfg["rate"], dev)
    fp8 = False if args.precision == "native" else args.precision
    psnr = lambda a, b: 10 * math.log10(((b.float().max() - b.float().min()).item() ** 2) /
                                        max(((a.float() - b.float()) ** 2).mean().item(), 1e-30))
    S = latent[0] * latent[1] * latent[2]
    print(f"# routed vs native attention on structured inputs: {args.config} latent {latent} S={S} tile {tile} window {window} "
          f"coreset {group} r={cfg['rate']}, {H} heads, bf16 inputs, routed precision {args.precision}; video tokens only (wan form)")
    print(f"# {torch.cuda.get_device_properties(0).name}; PSNR over the data range of the native output, dB")
    print("mix            heads(f/c/s)  structure  noise   whole-op   full  coreset  sliding-tile")
    for mix in ("uniform", "sparse-heavy", "all-lowres", "all-sliding"):
        experts = [int(e) for e in bench.layer_experts(dict(heads=H), mix, 0)]
        route = HeadRouting.from_expert_ids(experts, dev)
        for matched in (True, False):
            for noise in (0.0, 0.05, 0.1, 0.25, 0.5, 1.0):
                gen = torch.Generator(device=dev).manual_seed(123)
                q, k, v = (x.to(torch.bfloat16) for x in structured_layer(latent, experts, tile, group, noise, gen, dev, matched=matched))
                out = routed_attention(q, k, v, route, geom, model="wan", fp8=fp8)
                ref = dense_attention(q, k, v)
                per = []
                for e in range(3):
                    hs = [h for h in range(H) if experts[h] == e]
                    per.append(f"{psnr(out[0, hs], ref[0, hs]):7.1f}" if hs else "      -")
                n = [experts.count(e) for e in range(3)]
                print(f"{mix:14s} {n[0]:2d}/{n[1]:2d}/{n[2]:2d}       {'matched' if matched else 'SWAPPED'}   {noise:5.2f}   "
                      f"{psnr(out, ref):8.2f} {per[0]} {per[1]}  {per[2]}", flush=True)
                del q, k, v, out, ref
    print("# matched: sliding-tile heads LOCAL (Gaussian neighbourhood of a third of a tile, 16 logits above the background), coreset "
          "heads REDUNDANT (q, k, v constant over each coreset window), full-attention heads white noise; SWAPPED: the two "
          "structures exchanged (a router that chose wrongly); noise = share of white noise mixed into every tensor (1.00 = the "
          "white-noise floor of the method).")


if __name__ == "__main__":
    main()
