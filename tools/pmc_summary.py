#!/usr/bin/env python
"""Per-kernel sums of rocprofv3 --pmc passes (counter_collection.csv files found under the given directories).

    python tools/pmc_summary.py gpurun_out/r2/pmc_a gpurun_out/r2/pmc_b [--match attn8] [--json out.json]

For every kernel whose name contains --match: launches, average duration (from the counter file's timestamps), each
counter's value per launch, and the derived figures used in DESIGN.md (effective clock = GRBM_GUI_ACTIVE / 8 /
duration; MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x elapsed shader cycles))."""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--match", default="attn")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    per = defaultdict(lambda: dict(counters=defaultdict(float), launches=defaultdict(set), dur=defaultdict(dict)))
    for d in a.dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                name = row["Kernel_Name"]
                if a.match not in name:
                    continue
                k = per[name]
                c = row["Counter_Name"]
                k["counters"][c] += float(row["Counter_Value"])
                k["launches"][c].add((f, row["Dispatch_Id"]))
                k["dur"][c][(f, row["Dispatch_Id"])] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    out = {}
    for name, k in per.items():
        rec = {}
        for c, v in k["counters"].items():
            n = len(k["launches"][c])
            rec[c] = dict(per_launch=v / n, launches=n, avg_duration_ms=sum(k["dur"][c].values()) / n / 1e6)
        d = {}
        if "GRBM_GUI_ACTIVE" in rec:
            g = rec["GRBM_GUI_ACTIVE"]
            d["effective_clock_ghz"] = g["per_launch"] / 8 / (g["avg_duration_ms"] * 1e6)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in rec and "effective_clock_ghz" in d:
            m = rec["SQ_VALU_MFMA_BUSY_CYCLES"]
            cyc = m["avg_duration_ms"] * 1e6 * d["effective_clock_ghz"]
            d["mfma_pipe_utilisation"] = m["per_launch"] / (1024 * cyc)
        if "FETCH_SIZE" in rec or "WRITE_SIZE" in rec:
            # MI355X_MICROARCH.md HBM section: FETCH_SIZE (KiB) counts half the bytes of wide streaming reads on gfx950
            fb = rec.get("FETCH_SIZE", {}).get("per_launch", 0.0) * 1024 * 2
            wb = rec.get("WRITE_SIZE", {}).get("per_launch", 0.0) * 1024
            d["hbm_bytes_per_launch"] = fb + wb
        out[name] = dict(counters=rec, derived=d)
    txt = json.dumps(out, indent=1)
    print(txt)
    if a.json:
        open(a.json, "w").write(txt)


if __name__ == "__main__":
    main()
