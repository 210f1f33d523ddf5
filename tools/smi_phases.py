#!/usr/bin/env python
"""Join the [t0 .. t1] window each probe line prints with rocm-smi samples taken beside the run (round 4: tools/measure/r4_probe_shape.sh in the history of the tree):
mean socket power and sclk over the second half of each window."""
import json
import re
import sys


def main(probe, samples):
    rows, t = [], None
    for ln in open(samples):
        ln = ln.strip()
        if ln.startswith("t "):
            t = float(ln[2:])
        elif ln.startswith("{") and t is not None:
            try:
                card = next(iter(json.loads(ln).values()))
            except Exception:  # noqa: BLE001
                continue
            pw = next((float(v) for k, v in card.items() if "Power" in k and "W" in k), None)
            sclk = next((v for k, v in card.items() if k.startswith("sclk")), None)
            if isinstance(sclk, str):
                sclk = float("".join(c for c in sclk.strip("()").lower().replace("mhz", "") if c.isdigit() or c == "."))
            rows.append((t, pw, sclk))
    for ln in open(probe):
        m = re.search(r"\[(\d+\.\d+) \.\. (\d+\.\d+)\]", ln)
        if not m:
            print(ln.rstrip())
            continue
        t0, t1 = float(m.group(1)), float(m.group(2))
        mid = 0.5 * (t0 + t1)
        sel = [(p, c) for (t, p, c) in rows if mid <= t <= t1 and p is not None]
        pw = sum(p for p, _ in sel) / len(sel) if sel else float("nan")
        ck = [c for _, c in sel if c]
        print(f"{ln[:ln.index('[')].rstrip()}  smi: {pw:6.0f} W  sclk {sum(ck) / len(ck) if ck else float('nan'):5.0f} MHz  ({len(sel)} samples)")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
