#!/usr/bin/env python
"""Is the attention loop inside the chip's power envelope?  Samples the driver's gpu_metrics (rocm-smi: socket power, shader
clock) while each of these runs back to back for a few seconds: a library GEMM in fp16 and in e4m3 (hipBLASLt through
torch), the dense 16-bit attention launch, the dense e4m3 attention launch.  Prints power, clock and TFLOP/s per phase."""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def sample():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=20).stdout
        card = next(iter(json.loads(out).values()))
        pw = next((float(v) for k, v in card.items() if "Power" in k and "W" in k), None)
        sclk = next((v for k, v in card.items() if k.startswith("sclk")), None)
        if isinstance(sclk, str):
            sclk = float("".join(c for c in sclk.strip("()").lower().replace("mhz", "") if c.isdigit() or c == "."))
        return pw, sclk, card
    except Exception:  # noqa: BLE001
        return None, None, None


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.rows, self.stop, self.phase = [], False, "idle"

    def run(self):
        while not self.stop:
            p, c, _ = sample()
            self.rows.append((self.phase, p, c))
            time.sleep(0.03)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--S", type=int, default=75600)
    ap.add_argument("--H", type=int, default=8)
    args = ap.parse_args()
    from vorta_amd import ops
    dev = torch.device("cuda:0")
    print("raw sample:", json.dumps(sample()[2])[:600], flush=True)
    n = 8192
    a16, b16 = (torch.randn((n, n), device=dev).to(torch.float16) for _ in range(2))
    phases = {"gemm fp16 8192^3": (lambda: a16 @ b16, 2.0 * n ** 3)}
    try:
        a8, b8 = a16.to(torch.float8_e4m3fn), b16.to(torch.float8_e4m3fn).t().contiguous().t()
        one = torch.ones((), device=dev)
        torch._scaled_mm(a8, b8, scale_a=one, scale_b=one, out_dtype=torch.bfloat16)
        phases["gemm e4m3 8192^3 (torch._scaled_mm)"] = (
            lambda: torch._scaled_mm(a8, b8, scale_a=one, scale_b=one, out_dtype=torch.bfloat16), 2.0 * n ** 3)
    except Exception as e:  # noqa: BLE001
        print("no e4m3 library GEMM here:", repr(e)[:200], flush=True)
    S, H = args.S, args.H
    q, k, v = (torch.randn((H, S, 128), device=dev, dtype=torch.float16) for _ in range(3))
    o = torch.empty_like(q)
    f8 = ops.fp8_quantize_qkv(q, k, v, center_k=True)
    fl = 4.0 * S * S * 128 * H
    phases[f"attention fp16 dense S={S} H={H}"] = (lambda: ops.attn_fwd(q, k, v, o, n_q=S, n_kv=S), fl)
    phases[f"attention e4m3 dense S={S} H={H}"] = (
        lambda: ops.attn_fwd(f8.q, f8.k, f8.v, o, n_q=S, n_kv=S, v_descale=f8.v_descale), fl)
    sm = Sampler()
    sm.start()
    time.sleep(1.0)
    res = {}
    for name, (fn, flops) in phases.items():
        fn(); torch.cuda.synchronize()
        sm.phase = name
        t0 = time.time(); it = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        while time.time() - t0 < args.seconds:
            for _ in range(4):
                fn()
            it += 4
            torch.cuda.synchronize()
        e1.record(); torch.cuda.synchronize()
        res[name] = flops * it / (e0.elapsed_time(e1) * 1e9)
        sm.phase = "idle"
        time.sleep(1.0)
    sm.stop = True
    sm.join()
    for name in ["idle"] + list(phases):
        rows = [(p, c) for ph, p, c in sm.rows if ph == name and p is not None]
        rows = rows[len(rows) // 4:] if name != "idle" else rows  # drop the ramp
        if not rows:
            print(f"{name:44s}: no samples"); continue
        pw = sum(r[0] for r in rows) / len(rows)
        cl = [r[1] for r in rows if r[1]]
        print(f"{name:44s}: {pw:7.0f} W  sclk {sum(cl) / max(len(cl), 1):6.0f} MHz  "
              + (f"{res[name]:7.0f} TFLOP/s" if name in res else "") + f"  ({len(rows)} samples)", flush=True)


if __name__ == "__main__":
    main()
