"""Child process of tests/test_hip_rccl_single_rank.py: ONE rank with a REAL RCCL process group (backend "nccl").

The one-GPU box cannot hold two RCCL ranks (RCCL refuses two ranks on one device: "invalid usage"), and the gloo
rehearsals take the host-staged branch of the exchange.  This run takes the DIRECT branch -- the calls the driver's 8-GPU
run makes: `all_to_all_single(..., async_op=True)` with equal and with per-rank splits, for the whole tensor and per slot
group (round 6: the only transport), the MAX / SUM all-reduces of the 8-bit paths' statistics, the text all-gather -- on a
world of one, where every collective is a copy RCCL performs on its own
stream.  What it proves: the dtypes and shapes are ones RCCL accepts, the handles are waited for where the consumers need the
data (a missing wait shows as a mismatch against the loopback run, which does the same copies in stream order), and the
self-check passes on this transport.  What it cannot prove: anything about peers."""
import datetime
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from vorta_amd import ulysses
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29683")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=60))
    cfg = dict(bench.CONFIGS["tiny"])
    L = cfg["layers"]
    layer_ids = [bench.layer_experts(cfg, "uniform", l) for l in range(L)]
    per_head = bench.algorithmic_flops(cfg, layer_ids[0])[1]
    report = {"backend": dist.get_backend(), "cases": []}
    calls = {}

    def counted(name):  # which collectives of torch.distributed the exchange really issued on this transport
        fn = getattr(dist, name)

        def wrapper(*a, **kw):
            calls[name] = calls.get(name, 0) + 1
            return fn(*a, **kw)
        setattr(dist, name, wrapper)
    for name in ("all_to_all_single", "all_reduce", "all_gather", "all_gather_into_tensor", "batch_isend_irecv"):
        counted(name)
    ok = True
    for prec, dt in ((False, torch.float16), (False, torch.bfloat16), (True, torch.bfloat16), ("fp8pv", torch.bfloat16),
                     ("i8pv", torch.bfloat16)):
        for placement, groups in (("even", 1), ("even", 2), ("uneven", 2), ("split", 1)):
            outs = {}
            a2a_before = calls.get("all_to_all_single", 0)
            for loopback in (True, False):
                sp = ulysses.UlyssesRoutedAttention(cfg, layer_ids, per_head, dev, dt, 0, 1, groups=groups, loopback=loopback,
                                                    fp8=prec, placement=placement)
                sc = sp.selfcheck(0) if not loopback else {"ok": True}
                res = []
                for l in range(L):
                    sp.layer(l)
                    res.append((sp.out_shard.clone(), None if sp.out_text is None else sp.out_text.clone()))
                torch.cuda.synchronize()
                outs[loopback] = (res, sc)
            same = all(torch.equal(a[0], b[0]) and (a[1] is None or torch.equal(a[1], b[1]))
                       for a, b in zip(outs[True][0], outs[False][0]))
            finite = all(bool(torch.isfinite(a[0].float()).all()) for a in outs[False][0])
            case = {"precision": str(prec), "dtype": str(dt), "placement": placement, "groups": groups,
                    "selfcheck_ok": bool(outs[False][1]["ok"]), "failed": outs[False][1].get("failed", []),
                    "equals_loopback": bool(same), "finite": finite,
                    "a2a_calls": calls.get("all_to_all_single", 0) - a2a_before}
            ok = ok and case["selfcheck_ok"] and same and finite
            report["cases"].append(case)
    dist.barrier()
    dist.destroy_process_group()
    report["ok"] = ok
    report["collective_calls"] = calls
    print(json.dumps(report), flush=True)
    sys.exit(0 if ok else 4)


if __name__ == "__main__":
    main()
