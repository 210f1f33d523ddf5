#!/usr/bin/env python
"""DRAM-side view of the fused attention launch.  rocprofv3's FETCH_SIZE / WRITE_SIZE count what leaves the L2s (requests on
the TCC -> EA interface); the 256 MiB Infinity Cache sits behind that interface and no counter rocprofv3 lists on this
image (rocprofv3 -L: profiles/r03_counter_list_memory_side.txt) tells its hits from HBM accesses.  The memory controllers'
own activity is in the driver's gpu_metrics table (`rocm-smi --showmemuse`: "GPU Memory Read/Write Activity (%)" = UMC
activity).  This script samples it while (A) a streaming copy of known bandwidth runs -- the calibration: percent per
TB/s -- and (B) the fused routed-attention layers of bench.py's headline workload run back to back, and converts (B)'s
activity into HBM bytes per launch.  Coarse (the table is a ~1 ms moving average sampled a few times per second), but it
is a DRAM-side number."""
import argparse
import json
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def sample():
    try:
        out = subprocess.run(["rocm-smi", "--showmemuse", "--showuse", "--json"], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out)
        card = next(iter(d.values()))
        mem = next((float(v) for k, v in card.items() if "Read/Write Activity" in k), None)
        acc = next((float(v) for k, v in card.items() if k.strip() == "Memory Activity"), None)  # accumulated counter
        return mem, acc
    except Exception as e:  # noqa: BLE001
        return None, None


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.rows, self.stop, self.phase = [], False, "idle"

    def run(self):
        while not self.stop:
            m, u = sample()
            self.rows.append((time.time(), self.phase, m, u))
            time.sleep(0.05)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="hunyuan-129f")
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--seconds", type=float, default=8.0)
    ap.add_argument("--json", default=None)
    args = ap.parse_args()
    import bench as B
    from vorta_amd import ops
    from vorta_amd.routed import HeadRouting, RoutedGeometry, routed_attention
    dev = torch.device("cuda:0")
    cfg = dict(B.CONFIGS[args.config], dtype=args.dtype)
    fp8 = args.dtype == "fp8"
    dt = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    H, L, T, te = cfg["heads"], cfg["layers"], cfg["text"], cfg["text_valid"]
    S = cfg["latent"][0] * cfg["latent"][1] * cfg["latent"][2]
    geom = RoutedGeometry(cfg["latent"], cfg["tile"], cfg["window"], cfg["group"], cfg["rate"], dev)
    routings = [HeadRouting.from_expert_ids(B.layer_experts(cfg, "uniform", l), dev) for l in range(L)]
    sets = [tuple(torch.randn((1, H, S + T, 128), device=dev, dtype=dt) for _ in range(3)) for _ in range(2)]
    out = torch.empty_like(sets[0][0])
    f8 = ops.fp8_quantize_qkv(*(x[0] for x in sets[0])) if fp8 else None
    hy = cfg["model"] == "hunyuan"

    def layers(n):
        for l in range(n):
            q, k, v = sets[l % 2]
            routed_attention(q, k, v, routings[l % L], geom, model=cfg["model"], text_len=T, text_valid=te, out=out, fp8=fp8,
                             fp8_operands=f8)

    layers(4)
    torch.cuda.synchronize()
    src = torch.empty(2 << 30, dtype=torch.uint8, device=dev)
    dst = torch.empty_like(src)
    sm = Sampler()
    sm.start()
    time.sleep(1.0)
    res = {}
    # (A) calibration: device-to-device copy, 2 GiB read + 2 GiB written per call
    sm.phase = "copy"
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    t_end = time.time() + args.seconds
    e0.record()
    while time.time() < t_end:
        for _ in range(20):
            dst.copy_(src)
        n += 20
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    copy_tbs = n * 2 * src.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e12
    sm.phase = "idle"
    time.sleep(1.0)
    # (A') a second calibration point near the rate in question: the same copy at a ~6 % duty cycle
    sm.phase = "slowcopy"
    small_s, small_d = src[:256 << 20], dst[:256 << 20]
    n = 0
    t0 = time.time()
    t_end = t0 + args.seconds
    while time.time() < t_end:
        small_d.copy_(small_s)
        torch.cuda.synchronize()
        n += 1
        time.sleep(0.0017)
    slow_tbs = n * 2 * small_s.numel() / (time.time() - t0) / 1e12
    sm.phase = "idle"
    time.sleep(1.0)
    # (B) the fused layer launches back to back
    sm.phase = "attention"
    n = 0
    t_end = time.time() + args.seconds
    e0.record()
    while time.time() < t_end:
        layers(10)
        n += 10
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    ms_per_layer = e0.elapsed_time(e1) / n
    sm.phase = "idle"
    time.sleep(0.5)
    sm.stop = True
    sm.join(timeout=30)

    def stats(phase):
        v = [m for _, p, m, _ in sm.rows if p == phase and m is not None]
        v = v[1:-1] if len(v) > 4 else v  # drop the edges of the phase
        return (sum(v) / len(v), min(v), max(v), len(v)) if v else (None, None, None, 0)

    def acc_rate(phase):
        """accumulated-activity counter per second over the phase (finer than the integer percent)"""
        v = [(t, a) for t, p, _, a in sm.rows if p == phase and a is not None]
        v = v[1:-1] if len(v) > 4 else v
        return (v[-1][1] - v[0][1]) / (v[-1][0] - v[0][0]) if len(v) >= 2 and v[-1][0] > v[0][0] else None

    a, b, i, sl = stats("copy"), stats("attention"), stats("idle"), stats("slowcopy")
    ra, rb, ri, rs = acc_rate("copy"), acc_rate("attention"), acc_rate("idle"), acc_rate("slowcopy")
    res = {"workload": f"{args.config} uniform {args.dtype}", "copy_TBps": round(copy_tbs, 3), "slow_copy_TBps": round(slow_tbs, 3),
           "umc_activity_percent": {"copy": a, "slowcopy": sl, "attention": b, "idle": i},
           "memory_activity_counter_per_s": {"copy": ra, "slowcopy": rs, "attention": rb, "idle": ri},
           "attention_ms_per_layer": round(ms_per_layer, 3),
           "algorithmic_min_bytes_per_layer": (S + T) * H * 128 * ((3 if fp8 else 6) + 2)}
    if a[0] and b[0] is not None:
        per_pct = copy_tbs / a[0]  # TB/s of HBM traffic per percent of UMC activity
        bw = b[0] * per_pct
        res["from_percent"] = {"attention_hbm_TBps": round(bw, 3), "bytes_per_layer": round(bw * 1e12 * ms_per_layer * 1e-3),
                               "slowcopy_check_TBps": round(sl[0] * per_pct, 3) if sl[0] is not None else None}
    if ra and rb is not None and ri is not None and ra > ri:
        per = copy_tbs / (ra - ri)
        bw = max(rb - ri, 0.0) * per
        res["from_counter"] = {"attention_hbm_TBps": round(bw, 3), "bytes_per_layer": round(bw * 1e12 * ms_per_layer * 1e-3),
                               "slowcopy_check_TBps": round(max(rs - ri, 0.0) * per, 3) if rs is not None else None}
    print(json.dumps(res))
    if args.json:
        json.dump(res, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
