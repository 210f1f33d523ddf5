#!/usr/bin/env python
"""Diagnostic: where the waves of the int8-score attention loop spend their cycles.  Needs the -DVORTA_TRACE_I8=i builds
(i = 1 ... 7; one interval per build), e.g.
    bash tools/dbg/build_i8_variants.sh tri1 "-DVORTA_I8_DIAG -DVORTA_TRACE_I8=1" ... tri7 "-DVORTA_I8_DIAG -DVORTA_TRACE_I8=7"
    python tools/trace_i8.py            # runs itself once per library (child processes)
Interval i = shader cycles per step between stamps i-1 and i:
  0 step start | 1 after the tile requests at the top (role Y's) | 2 before the matrix part | 3 after it | 4 after the requests
  behind it (role X's) | 5 before the end-of-step wait (a stamp right behind the last MFMA's issue reads garbage: interval 5
  follows from the others) | 6 before the barrier | 7 after it"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = ["", "tile requests (role Y)", "VALU part (role Y)", "matrix part", "tile requests (role X)", "VALU part (role X)",
         "vmcnt/lgkmcnt wait", "barrier"]


def one(i):
    import torch
    from vorta_amd import _C, ops
    S, H = int(os.environ.get("S", 75600)), int(os.environ.get("H", 8))
    dev = torch.device("cuda:0")
    q, k, v = (torch.randn((H, S, 128), device=dev, dtype=torch.bfloat16) for _ in range(3))
    o = torch.empty_like(q)
    v8, vd, _ = ops.fp8_quantize_v(v)
    i8 = ops.i8_quantize_k(q, k)
    br, nw = 256, 8
    n_wg = H * ((S + br - 1) // br)
    tr = torch.zeros((n_wg, nw, 2), dtype=torch.int32, device=dev)
    a, keep = ops._attn_args(q, i8.k8, v8, o, n_q=S, n_kv=S, block_rows=br, v_descale=vd, i8=i8)
    a.ws_ml = tr.data_ptr()
    lib, ext = _C.lib(), a._ext

    def launch():
        _C.check(lib.vorta_attn_fwd_i8(C.byref(a), C.byref(ext), ops._stream()), "vorta_attn_fwd_i8")

    launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    t = tr.cpu().to(torch.int64) & 0xffffffff
    per = t[..., 0].double() / t[..., 1].double().clamp(min=1)
    x, y = per[:, :4].mean().item(), per[:, 4:].mean().item()
    print(f"interval {i} {NAMES[i]:32s}: role X {x:7.1f}  role Y {y:7.1f} cycles/step   "
          f"({ms:.3f} ms, {4.0 * S * S * 128 * H / ms / 1e9:.0f} TFLOP/s, steps {t[0, 0, 1].item()})", flush=True)
    print("           per wave 0..7 (wave 0 also requests the key-bias piece): " + " ".join(f"{per[:, w].mean().item():7.1f}" for w in range(8)), flush=True)


def main():
    if len(sys.argv) > 1:
        return one(int(sys.argv[1]))
    for i in (1, 2, 3, 4, 5, 6, 7):
        lib = os.path.join(ROOT, "vorta_amd", "csrc", f"libvorta_hip_tri{i}.so")
        if not os.path.exists(lib):
            print("missing", lib)
            continue
        subprocess.run([sys.executable, os.path.abspath(__file__), str(i)], env=dict(os.environ, VORTA_HIP_LIB=lib))


if __name__ == "__main__":
    main()
