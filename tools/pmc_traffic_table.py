#!/usr/bin/env python
"""Fold the per-workload FETCH_SIZE / WRITE_SIZE passes of tools/measure_r3_pmc.sh into one table keyed by bench.py's
workload string ("<config> <mix> <dtype>"): per attention kernel, bytes leaving the L2s per launch (FETCH_SIZE x 1024 x 2 --
gfx950 tallies 64 of every 128 streamed bytes, MI355X_MICROARCH.md 'HBM' -- + WRITE_SIZE x 1024), the algorithmic minimum
(read Q,K,V + write O once) and their ratio.  bench.py reads the table for `roofline.traffic`."""
import argparse
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def kernel_sums(d):
    per = defaultdict(lambda: dict(v=0.0, ids=set(), dur=0))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            if "attn" not in name or "combine" in name:
                continue
            k = per[name.split("(")[0].replace("void (anonymous namespace)::", "").strip()]
            k["v"] += float(row["Counter_Value"])
            key = (f, row["Dispatch_Id"])
            if key not in k["ids"]:
                k["ids"].add(key)
                k["dur"] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    return {n: dict(per_launch=k["v"] / len(k["ids"]), launches=len(k["ids"]), avg_ms=k["dur"] / len(k["ids"]) / 1e6)
            for n, k in per.items() if k["ids"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--json", required=True)
    ap.add_argument("--head", default=os.environ.get("VORTA_TREE_HEAD"), help="git head of the profiled tree (the GPU box has no .git: "
                    "the caller passes it, e.g. VORTA_TREE_HEAD=$(git rev-parse --short=12 HEAD) in the gpurun command)")
    ap.add_argument("--source", default="tools/measure/r6_gpu.sh")
    a = ap.parse_args()
    import bench as B
    table = {}
    for d in sorted(glob.glob(os.path.join(a.root, "*_*_*"))):
        if not os.path.isdir(d):
            continue
        cfg_name, mix, dt = os.path.basename(d).rsplit("_", 2)
        cfg = B.CONFIGS[cfg_name]
        S = cfg["latent"][0] * cfg["latent"][1] * cfg["latent"][2]
        esz_in = 1 if dt in ("fp8", "i8pv") else 2  # fp8pv: q, k 16-bit, v e4m3 (counted as 16-bit: an upper minimum)
        # the attention launch reads q,k,v (e4m3 copies under fp8) and writes a 16-bit output
        minimum = (S + cfg["text"]) * cfg["heads"] * 128 * (3 * esz_in + 2)
        fetch, write = kernel_sums(os.path.join(d, "FETCH_SIZE")), kernel_sums(os.path.join(d, "WRITE_SIZE"))
        kernels = {}
        for name in sorted(set(fetch) | set(write)):
            f, w = fetch.get(name), write.get(name)
            fb = (f["per_launch"] if f else 0.0) * 1024 * 2
            wb = (w["per_launch"] if w else 0.0) * 1024
            ms = (f or w)["avg_ms"]
            kernels[name] = {"launches": (f or w)["launches"], "avg_duration_ms": round(ms, 3),
                             "FETCH_SIZE_KiB_per_launch": round(f["per_launch"]) if f else None,
                             "WRITE_SIZE_KiB_per_launch": round(w["per_launch"]) if w else None,
                             "l2_miss_bytes_per_launch": round(fb + wb), "rate_TB_per_s": round((fb + wb) / (ms * 1e-3) / 1e12, 3)}
        table[f"{cfg_name} {mix} {dt}"] = {"kernels": kernels, "algorithmic_min_bytes_per_fused_launch": minimum}
    out = {"what": "bytes leaving the L2s (TCC -> EA requests: FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024) per launch of the "
                   "attention kernels; the Infinity Cache sits BEHIND this interface (its hits are counted here), the "
                   "DRAM-side estimate is profiles/r0N_umc_activity_*.json",
           "head": a.head,
           "source": a.source + ": rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, no trace "
                     "domains, python3 bench.py --config C --mix M --dtype D --steps 1 --warmup 0",
           "workloads": table}
    json.dump(out, open(a.json, "w"), indent=1)
    for wl, t in table.items():
        for k, v in t["kernels"].items():
            print(f"{wl:34s} {k:48s} {v['l2_miss_bytes_per_launch'] / 1e9:8.2f} GB / launch  {v['avg_duration_ms']:8.2f} ms  x{v['launches']}")


if __name__ == "__main__":
    main()
