#!/usr/bin/env python
"""Ad-hoc probe: L2->fabric fetch traffic of a routed layer, fused grid vs one launch per expert, for different
expert mixes.  Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vorta_amd.routed import HeadRouting, RoutedGeometry, routed_attention

dev = torch.device("cuda:0")
latent, T, te, H = (33, 45, 80), 256, 96, 24
S = latent[0] * latent[1] * latent[2]
q, k, v = (torch.randn((1, H, S + T, 128), device=dev, dtype=torch.float16) for _ in range(3))
o = torch.empty_like(q)
geom = RoutedGeometry(latent, (11, 9, 8), (3, 3, 3), (3, 3, 2), 0.5, dev)
geom.sta_tables(te)
mixes = {"full+lowres": [0] * 8 + [1] * 8, "full+sliding": [0] * 8 + [2] * 8, "lowres+sliding": [1] * 8 + [2] * 8,
         "all3": [0] * 8 + [1] * 8 + [2] * 8}
for name, ex in mixes.items():
    ex = ex + [0] * 0
    heads = list(range(len(ex)))
    r = HeadRouting.from_expert_ids(ex + [9] * (H - len(ex)), dev)  # expert 9 = unused heads
    for fused in (True, False):
        routed_attention(q, k, v, r, geom, model="hunyuan", text_len=T, text_valid=te, out=o, fused=fused,
                         sliding_block_rows=256)
        torch.cuda.synchronize()
        print(name, "fused" if fused else "serial", flush=True)
