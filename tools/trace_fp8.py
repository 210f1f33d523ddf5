#!/usr/bin/env python
"""Diagnostic: where the waves of the e4m3 attention loop spend their cycles.  Needs the -DVORTA_TRACE8=i builds
(i = 1, 2, 4, 5; one interval per build: the loop has no registers to spare), e.g.
    for i in 1 2 4 5; do VORTA_BUILD_SUFFIX=_tr8$i VORTA_EXTRA_FLAGS="-DVORTA_FP8_DIAG -DVORTA_TRACE8=$i" python -m vorta_amd.build; done
    python tools/trace_fp8.py            # runs itself once per library (child processes)
Interval i = shader cycles per step between stamps i-1 and i:
  0 step start | 1 before the matrix part | 2 after it | 3 before the end-of-step wait | 4 before the barrier | 5 after it"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = ["", "DMA issue + first VALU part (role Y)", "matrix part", "second VALU part (role X)", "vmcnt/lgkmcnt wait",
         "barrier"]


def one(i):
    import torch
    from vorta_amd import _C, ops
    S, H = int(os.environ.get("S", 75600)), int(os.environ.get("H", 8))
    dev = torch.device("cuda:0")
    q, k, v = (torch.randn((H, S, 128), device=dev, dtype=torch.bfloat16) for _ in range(3))
    o = torch.empty_like(q)
    # TRACE_SCALE_MUL: multiplier on the softmax scale (8 for the byte-direct pack experiment of
    # profiles/r03_fp8_loop_experiments.txt, whose q8 . k8 carried 8 x the score; that build is not in the tree)
    f8 = ops.fp8_quantize_qkv(q, k, v, scale=float(os.environ.get("TRACE_SCALE_MUL", 1)) / 128 ** 0.5)
    br, nw = 256, 8
    n_wg = H * ((S + br - 1) // br)
    tr = torch.zeros((n_wg, nw, 2), dtype=torch.int32, device=dev)
    a, keep = ops._attn_args(f8.q, f8.k, f8.v, o, n_q=S, n_kv=S, block_rows=br, v_descale=f8.v_descale)
    a.ws_ml = tr.data_ptr()
    lib, ext = _C.lib(), a._ext

    def launch():
        _C.check(lib.vorta_attn_fwd_fp8(C.byref(a), C.byref(ext), ops._stream()), "vorta_attn_fwd_fp8")

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
    print(f"interval {i} {NAMES[i]:38s}: role X {x:7.1f}  role Y {y:7.1f} cycles/step   "
          f"({ms:.3f} ms, {4.0 * S * S * 128 * H / ms / 1e9:.0f} TFLOP/s, steps {t[0, 0, 1].item()})", flush=True)


def main():
    if len(sys.argv) > 1:
        return one(int(sys.argv[1]))
    # interval 3 (the second VALU part) is not stamped: a stamp right behind the last MFMA's issue reads garbage here;
    # it follows from the others, because the waves of both roles run the same number of cycles per step
    for i in (1, 2, 4, 5):
        lib = os.path.join(ROOT, "vorta_amd", "csrc", f"libvorta_hip_tr8{i}.so")
        if not os.path.exists(lib):
            print("missing", lib)
            continue
        subprocess.run([sys.executable, os.path.abspath(__file__), str(i)], env=dict(os.environ, VORTA_HIP_LIB=lib))
    print("step = sum of role Y's intervals (its interval 3 is empty); role X's second VALU part = step - its other four")


if __name__ == "__main__":
    main()
