"""The direct (RCCL) branch of the sequence-parallel exchange on a world of ONE rank: tests/_rccl_single_rank.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exchange_on_a_real_rccl_group_of_one_rank_equals_the_loopback_run():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29683", HSA_ENABLE_IPC_MODE_LEGACY="0", VORTA_SP_FORCE_COLLECTIVES="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_single_rank.py")], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, (r.stdout[-1500:], r.stderr[-3000:])
    rep = json.loads(lines[-1])
    bad = [c for c in rep["cases"] if not (c["selfcheck_ok"] and c["equals_loopback"] and c["finite"])]
    assert rep["backend"] == "nccl" and len(rep["cases"]) == 20 and not bad and r.returncode == 0, (bad, r.stderr[-2000:])
    # the collectives went through torch.distributed (not around it): EVERY case -- the slot-group ones too, since round 6 a
    # slot group is a receive layout of its own -- issues q, k, v in and o back per layer and group as all_to_all_single
    # (+ the self-check's); nothing is exchanged point to point
    calls = rep["collective_calls"]
    assert calls.get("all_to_all_single", 0) >= 20 * 4 * 4 and calls.get("all_reduce", 0) > 0 and calls.get("all_gather", 0) > 0, calls
    assert calls.get("batch_isend_irecv", 0) == 0, calls
    grouped = [c for c in rep["cases"] if c["groups"] > 1]
    assert len(grouped) == 10 and all(c["a2a_calls"] >= 2 * 4 * 4 for c in grouped), grouped  # 2 groups x 4 layers x 4 tensors
    print(rep["collective_calls"])
