# The statistic behind precision "auto8": the root mean square of a head's int8 keys (vorta_i8_quantize_k: centred, channel-balanced,
# ONE scale per head = abs-max / 127) for every input family of tests/_fp8_inputs.py at full size, three heads each -- exact over all
# rows, and what vorta_i8_tail_flags sees over its ~1024 sampled rows (flag = rms < VORTA_I8_TAIL_MIN_RMS, default 3.2 counts).
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _fp8_inputs import NAMES, families  # noqa: E402
from vorta_amd import ops  # noqa: E402

dev = torch.device("cuda")
for geometry, latent, T in (("wan14b-81f", (21, 45, 80), 0), ("hunyuan-129f", (33, 45, 80), 256)):
    gen = torch.Generator(device=dev).manual_seed(1234)
    for key, q, k, v in families(latent, 3, T, gen, dev):
        q16, k16 = q.to(torch.bfloat16).contiguous(), k.to(torch.bfloat16).contiguous()
        i8 = ops.i8_quantize_k(q16, k16)
        k8 = i8.k8.float()
        rms = k8.pow(2).mean((1, 2)).sqrt().tolist()
        stride = max(1, k8.shape[1] // 1024) | 1
        rms_s = k8[:, ::stride].pow(2).mean((1, 2)).sqrt().tolist()
        flags = ops.i8_tail_flags(i8.k8).tolist()
        print(f"{geometry:13s} {NAMES[key][:58]:58s} rms of the int8 keys {', '.join(f'{x:5.2f}' for x in rms)}   sampled {', '.join(f'{x:5.2f}' for x in rms_s)}   flags {flags}")
