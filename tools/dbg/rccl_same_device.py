# Can two RCCL ranks share ONE GPU on this box?  (If yes, the direct "nccl" branch of the exchange can be rehearsed on the 1-GPU box;
# if RCCL refuses duplicate devices, only gloo rehearsals are possible.)  Round 5 on the pool's box: REFUSED -- both ranks get
# 'NCCL error ... invalid usage' from init; hence tests/test_hip_rccl_single_rank.py (one rank, collectives forced).  Run: python -m torch.distributed.run --nproc-per-node 2
#   --master-addr 127.0.0.1 --master-port 29571 tools/dbg/rccl_same_device.py
import datetime
import os
import sys

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", timeout=datetime.timedelta(seconds=60))
    x = torch.full((world * 4,), float(rank), device="cuda")
    y = torch.empty_like(x)
    dist.all_to_all_single(y, x)
    torch.cuda.synchronize()
    print(f"rank {rank}: all_to_all_single ok {y.tolist()}", flush=True)
    dist.destroy_process_group()
except Exception as e:  # noqa: BLE001
    print(f"rank {rank}: {type(e).__name__}: {str(e)[:400]}", flush=True)
    sys.exit(3)
