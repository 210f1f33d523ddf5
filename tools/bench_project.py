#!/usr/bin/env python
"""Steps 1-4 of a HunyuanVideo dual-stream block at the headline size (S = 118 800 video + 256 text tokens, hidden 3072,
24 heads): q/k/v projections, qk-norm, RoPE, text concat -- `HunyuanVideoFlashAttnProcessor._project` with the
projections written into one buffer per tensor (default) vs separate projections + torch.cat (VORTA_JOINT_PROJECTION=0)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn

from vorta_amd.attention import HunyuanVideoFlashAttnProcessor, hunyuan as hy


class Attn(nn.Module):
    def __init__(self, hidden, H, dtype, dev, dual):
        super().__init__()
        self.heads = H
        self.to_q, self.to_k, self.to_v = (nn.Linear(hidden, hidden) for _ in range(3))
        self.norm_q, self.norm_k = (nn.RMSNorm(128, eps=1e-6) for _ in range(2))
        self.add_q_proj = self.add_k_proj = self.add_v_proj = self.norm_added_q = self.norm_added_k = None
        if dual:
            self.add_q_proj, self.add_k_proj, self.add_v_proj = (nn.Linear(hidden, hidden) for _ in range(3))
            self.norm_added_q, self.norm_added_k = (nn.RMSNorm(128, eps=1e-6) for _ in range(2))
        self.to(dev).to(dtype)


def main():
    dev, dtype = torch.device("cuda:0"), torch.bfloat16
    S, T, hidden, H = int(os.environ.get("S", 118800)), 256, 3072, 24
    x = torch.randn((1, S, hidden), device=dev, dtype=dtype)
    e = torch.randn((1, T, hidden), device=dev, dtype=dtype)
    ang = torch.rand((S, 64), device=dev) * 6.28
    rope = (ang.cos().repeat_interleave(2, dim=1).contiguous(), ang.sin().repeat_interleave(2, dim=1).contiguous())
    proc = HunyuanVideoFlashAttnProcessor()
    for dual, joint in ((True, True), (True, False), (True, True), (True, False), (False, True), (False, False),
                        (False, True), (False, False)):
        attn = Attn(hidden, H, dtype, dev, dual)
        hy.JOINT_PROJECTION = joint
        with torch.no_grad():
            proc._project(attn, x, e, rope)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                proc._project(attn, x, e, rope)
            e1.record()
            torch.cuda.synchronize()
        print(f"_project, {'dual' if dual else 'single'}-stream block, S={S}+{T}: "
              f"{'one buffer per tensor' if joint else 'the reference route  '}: "
              f"{e0.elapsed_time(e1) / 5:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
