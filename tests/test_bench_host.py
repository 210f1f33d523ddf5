"""CPU: bench.py's multi-rank ORCHESTRATION -- the GPU-free supervisors of an N > 1 run (bench.py `supervise`) -- under the real
launcher (`python -m torch.distributed.run`, 2 ranks) with CPU stub workers (`VORTA_BENCH_STUB`: a gloo process group per attempt,
one all-reduce, one JSON line).  What the GPU rehearsals in tests/test_hip_bench.py cannot show quickly: a first attempt that
fails on every rank, on one rank only (the others left waiting in a collective), or never joins -- each must come back with
exit code 0 and ONE JSON line on stdout labelled `"fallback": "conservative"`; `--no-fallback` / `--conservative` leave a failed
attempt final."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(stub, extra=(), timeout_s="20"):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(VORTA_BENCH_STUB=stub, VORTA_BENCH_TIMEOUT_S=timeout_s, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "2", "--config", "tiny"] + list(extra)
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_supervised_run_relays_one_line():
    r = _run("ok")
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["metric"] == "stub" and lines[0]["value"] == 3.0 and "fallback" not in lines[0]
    assert "[stub] a line that is not JSON" in r.stderr and r.stdout.strip().count("\n") == 0  # stdout: the JSON line alone


@pytest.mark.parametrize("stub", ["fail", "fail0", "hang1"])
def test_failed_first_attempt_falls_back_to_one_conservative_child_per_rank(stub):
    """every rank fails / only rank 0 fails while rank 1 waits in the all-reduce / rank 1 never joins the rendezvous: the
    supervisors agree through the launcher's store, end the children that can only be waiting, and run --conservative"""
    r = _run(stub, timeout_s="30")
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout[-2000:]  # the failed attempt's JSON went to stderr
    j = lines[0]
    assert j["metric"] == "stub" and j["value"] == 3.0 and j["conservative"] is True and j["fallback"] == "conservative"
    assert "exit codes" in j["first_attempt_error"] and "first attempt failed" in r.stderr
    if stub != "hang1":
        assert "stub:" in j["first_attempt_error"] and "[bench attempt 0]" in r.stderr


def test_no_fallback_and_conservative_leave_a_failed_attempt_final():
    r = _run("fail", ["--no-fallback"])
    assert r.returncode != 0
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and "error" in lines[0], r.stdout[-1000:]
    r = _run("ok", ["--conservative"])
    assert r.returncode == 0 and _json_lines(r.stdout)[0]["conservative"] is True


def test_a_failed_teardown_does_not_throw_a_finished_measurement_away():
    """rank 1 leaves with a non-zero code AFTER rank 0 printed the line: the attempt counts (exit code 0, no fallback)"""
    r = _run("teardown1")
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["metric"] == "stub" and "fallback" not in lines[0] and lines[0]["conservative"] is False
    assert "teardown" in r.stderr and "first attempt failed" not in r.stderr
