#!/usr/bin/env python
"""Diagnostic (needs a -DVORTA_TRACE build passed through VORTA_HIP_LIB): per-wave cycles of the attention main loop
spent in the end-of-step wait (vmcnt/lgkmcnt) and inside s_barrier, for a dense launch."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vorta_amd import _C, ops


def main():
    S, H = int(os.environ.get("S", 32760)), int(os.environ.get("H", 12))
    dev = torch.device("cuda:0")
    q, k, v = (torch.randn((H, S, 128), device=dev, dtype=torch.bfloat16) for _ in range(3))
    o = torch.empty_like(q)
    br = int(os.environ.get("BLOCK_ROWS", 256))
    nw = br // 32
    n_wg = H * ((S + br - 1) // br)
    tr = torch.zeros((n_wg, nw, 8), dtype=torch.int64, device=dev)
    a, keep = ops._attn_args(q, k, v, o, n_q=S, n_kv=S, block_rows=br, variant=2)
    a.ws_ml = tr.data_ptr()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _C.check(_C.lib().vorta_attn_fwd(C.byref(a), ops._stream()), "vorta_attn_fwd")
    e0.record()
    for _ in range(4):
        _C.check(_C.lib().vorta_attn_fwd(C.byref(a), ops._stream()), "vorta_attn_fwd")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 4
    print(f"block_rows {br}: {ms:.3f} ms  {4.0 * S * S * 128 * H / ms / 1e9:.1f} TFLOP/s")
    if tr.abs().sum().item() == 0:
        return  # not a -DVORTA_TRACE build
    t = tr.double().cpu()
    tot, wait, bar, steps = t[..., 0], t[..., 1], t[..., 2], t[..., 3]
    mhz = (tot / t[..., 4]).mean().item() * 100.0  # wall_clock64 ticks at 100 MHz
    print(f"workgroups {n_wg}, steps/wave {steps.mean():.0f}, shader clock over the loops {mhz:.0f} MHz")
    print(f"loop cycles/step: {(tot / steps).mean():.0f}  (min {(tot / steps).min():.0f}, max {(tot / steps).max():.0f})")
    print(f"wait  cycles/step: {(wait / steps).mean():.0f}  = {100 * (wait / tot).mean():.1f} % of the loop")
    print(f"barrier cycles/step: {(bar / steps).mean():.0f}  = {100 * (bar / tot).mean():.1f} % of the loop")
    per_wave = (bar / tot).mean(0)
    print("barrier share by wave:", " ".join(f"{100 * x:.1f}" for x in per_wave.tolist()))
    per_wave = (wait / tot).mean(0)
    print("wait share by wave:   ", " ".join(f"{100 * x:.1f}" for x in per_wave.tolist()))
    # timeline per CU slot (last launch): entry -> end of loop of consecutive workgroups on the same CU
    ti = tr[:, 0].cpu()  # wave 0 of every workgroup
    start, end, hw = ti[:, 5], ti[:, 6], ti[:, 7]
    cu = (hw & 0xffff) >> 8 & 0xf
    se = (hw >> 13) & 0x7
    xcc = (hw >> 32) & 0xf
    key = (xcc * 8 + se) * 16 + cu
    t0 = start.min().item()
    span = (end.max().item() - t0) / 100.0
    busy = ((end - start).double().sum().item() / 100.0)
    slots = key.unique().numel() * (2 if br == 128 else 1)
    print(f"launch span {span:.0f} us over {key.unique().numel()} CUs; sum of (entry -> loop end) {busy:.0f} us "
          f"= {100 * busy / (span * slots):.1f} % of span x slots")
    gaps = []
    for k_ in key.unique().tolist():
        m = key == k_
        s_, e_ = start[m].sort().values, end[m].sort().values
        if br == 256 and len(s_) > 1:
            gaps += ((s_[1:] - e_[:-1]).double() / 100.0).tolist()
    if gaps:
        g = torch.tensor(gaps)
        print(f"gap loop-end -> next entry on the same CU: mean {g.mean():.1f} us, max {g.max():.1f} us, n={len(gaps)}")
    first = (start - t0).double() / 100.0
    print(f"first-round entries: within {first.sort().values[key.unique().numel() - 1]:.1f} us; last loop end at {span:.0f} us; "
          f"loop time per workgroup {((end - start).double() / 100.0).mean():.0f} us")


if __name__ == "__main__":
    main()
