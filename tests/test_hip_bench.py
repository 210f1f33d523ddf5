"""GPU: bench.py's contract -- one JSON line with the required keys -- at N=1 and through the N>1 code path
(2 ranks sharing the GPU, gloo host-staged rehearsal of the RCCL exchange), on the tiny configuration."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def _line(out: str) -> dict:
    lines = [l for l in out.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_single_gpu_line():
    r = subprocess.run([sys.executable, "bench.py", "--config", "tiny", "--steps", "2", "--warmup", "1"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r.stdout)
    assert KEYS <= set(j) and "cpu_baseline" in j
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["value"] > 0 and j["vs_baseline"] is None
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(j["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(j["cpu_baseline"]) and "workload" in j["config"]
    # the library kernels the reference borrows, beside ours (context fields; an error is named, never raised)
    assert "library_sdpa_tflops" in j["roofline"] and "step_ms_if_borrowed_routed" in j["config"], j["config"]
    assert j["config"]["step_ms_if_borrowed_routed"] is None or j["config"]["step_ms_if_borrowed_routed"] > 0


@pytest.mark.parametrize("dtype", ["fp16", "fp8"])
def test_bench_two_rank_rehearsal(dtype):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, VORTA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2",
                        "--config", "tiny", "--dtype", dtype, "--steps", "1", "--warmup", "1", "--no-cpu-baseline"], cwd=ROOT,
                       env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    j = _line(r.stdout)
    assert KEYS <= set(j) and j["n_gpus"] == 2 and j["scaling"] == "strong" and j["value"] > 0 and j["dtype"] == dtype
    # the pre-flight of the N > 1 run: tagged q,k,v through layer 0's exchange (v as e4m3 in the fp8 run), exact on every rank
    sc = j["exchange_selfcheck"]
    assert sc["ok"] is True and sc["bytes"] > 0 and sc["ms"] > 0 and sc["failed_on_this_rank"] == []
    assert ("v as e4m3" in sc["what"]) == (dtype == "fp8")
    # the N > 1 line explains itself: what the exchange costs alone, what the step hid, bytes per link against the link's peak
    ex = j["exchange"]
    assert {"exchange_ms_per_layer", "compute_ms_per_layer", "exposed_exchange_ms_per_layer", "bytes_per_link_per_layer",
            "bytes_per_link_per_16bit_tensor", "frac_of_link_peak", "frac_of_7_links", "egress_bytes_per_rank_per_layer"} <= set(ex)
    assert ex["exchange_ms_per_layer"] > 0 and ex["compute_ms_per_layer"] > 0 and ex["bytes_per_link_per_layer"] > 0
    # tiny: S = 1728, 2 ranks, 4 heads to the peer: 4 x 864 x 128 x 2 B per 16-bit tensor
    assert ex["bytes_per_link_per_16bit_tensor"] == 4 * 864 * 128 * 2 and ex["v_bytes_per_element_on_the_wire"] == (1 if dtype == "fp8" else 2)
    assert "fallback" not in j


def test_bench_selfcheck_catches_a_misordered_exchange_and_falls_back_to_a_conservative_child():
    """VORTA_SP_SELFCHECK_BREAK=1 (test-only): the last rank swaps two heads of its copy of the placement in the FIRST attempt.
    With --no-fallback the run stops before the warm-up with ONE JSON error line and a non-zero exit code.  By default every
    rank's GPU-free supervisor (bench.py `supervise`) starts one fresh --conservative child: exit code 0, ONE JSON line on
    stdout, labelled `"fallback": "conservative"` with the first attempt's error; the failed attempt's lines go to stderr.
    --conservative: auto placement, one slot group, v in 16 bits -- same layer output as the default exchange."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(VORTA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    base = [sys.executable, "bench.py", "--gpus", "2", "--config", "tiny", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(base + ["--no-fallback"], cwd=ROOT, env=dict(env, VORTA_SP_SELFCHECK_BREAK="1"), capture_output=True,
                       text=True, timeout=900)
    assert r.returncode != 0
    err = [l for l in r.stdout.splitlines() if l.startswith('{"error"')]
    assert len(err) == 1 and not [l for l in r.stdout.splitlines() if l.startswith('{"metric"')], r.stdout[-2000:]
    e = json.loads(err[0])
    assert e["exchange_selfcheck"]["ok"] is False and "self-check" in e["error"]
    # the same broken first attempt, fallback on (the default): the run comes back with a number, and says how
    r = subprocess.run(base, cwd=ROOT, env=dict(env, VORTA_SP_SELFCHECK_BREAK="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert len([l for l in r.stdout.splitlines() if l.startswith("{")]) == 1, r.stdout[-2000:]
    j = _line(r.stdout)
    assert j["fallback"] == "conservative" and "self-check" in j["first_attempt_error"] and j["exchange_selfcheck"]["ok"] is True
    assert "[bench attempt 0]" in r.stderr and "auto" not in j["config"]["parallelism"].split("head placement")[0][-8:]
    fps = {("fallback", "bf16"): j["output_fingerprint"]}
    for name, extra in (("default", []), ("conservative", ["--conservative"])):
        for dtype in ("bf16", "fp8"):
            r = subprocess.run(base + ["--dtype", dtype] + extra, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, (name, dtype, r.stdout[-1500:], r.stderr[-3000:])
            j = _line(r.stdout)
            assert j["exchange_selfcheck"]["ok"] is True and "fallback" not in j
            fps[(name, dtype)] = j["output_fingerprint"]
            if name == "conservative":
                assert "even head placement" in j["config"]["parallelism"]
                assert "v as e4m3" not in j["exchange_selfcheck"]["what"]
    assert fps[("default", "bf16")] == fps[("conservative", "bf16")] == fps[("fallback", "bf16")] != 0
    assert fps[("default", "fp8")] == fps[("conservative", "fp8")] != 0


def test_bench_plain_command_launches_its_ranks_and_fp8_exchange_variants_agree():
    """`python bench.py --gpus 2` with no launcher around it starts its two ranks itself (a child process running
    torch.distributed.run); the JSON names the ranks the process group saw.  fp8: the overlapped exchange (two slot
    groups, converted group by group) and v on the wire as e4m3 give the bytes of the plain exchange."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(VORTA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    lines = {}
    for name, extra in (("plain", ["--no-v-wire"]), ("vwire", []), ("groups", ["--sp-groups", "2"]),
                        ("groups16", ["--sp-groups", "2", "--no-v-wire"]),
                        ("uneven", ["--placement", "uneven"]), ("uneven_groups", ["--placement", "uneven", "--sp-groups", "2"])):
        r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--config", "tiny", "--dtype", "fp8", "--steps", "1",
                            "--warmup", "1", "--no-cpu-baseline"] + extra, cwd=ROOT, env=env, capture_output=True, text=True,
                           timeout=900)
        assert r.returncode == 0, (name, r.stdout[-1500:], r.stderr[-3000:])
        lines[name] = _line(r.stdout)
    j = lines["plain"]
    pg = j["process_group"]
    assert j["n_gpus"] == 2 and pg["world_size"] == 2 and pg["backend"] == "gloo" and len(pg["ranks"]) == 2
    assert sorted(r["rank"] for r in pg["ranks"]) == [0, 1] and len({r["pid"] for r in pg["ranks"]}) == 2
    fps = {name: l["output_fingerprint"] for name, l in lines.items()}
    # ... and so do ranks holding different numbers of heads (3 + 5 of the 8 heads, from the layers' routes)
    assert len(set(fps.values())) == 1 and fps["plain"] != 0, fps
    assert "uneven head placement" in lines["uneven"]["config"]["parallelism"]


def test_bench_fp8pv_lines():
    """--dtype fp8pv (scores in bf16, P V in e4m3): one GPU, and 2 ranks (v converted on the send side, exchanged as
    bytes) with one and two slot groups giving one fingerprint"""
    r = subprocess.run([sys.executable, "bench.py", "--config", "tiny", "--steps", "1", "--warmup", "1", "--dtype", "fp8pv",
                        "--no-gemm-ceiling", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r.stdout)
    assert j["dtype"] == "fp8pv" and j["roofline"]["kernel"].startswith("attn_mx_") and 3300 < j["roofline"]["peak"] < 3400
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(VORTA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    fps = []
    for extra in ([], ["--sp-groups", "2"]):
        r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--config", "tiny", "--dtype", "fp8pv", "--steps", "1",
                            "--warmup", "1", "--no-cpu-baseline"] + extra, cwd=ROOT, env=env, capture_output=True, text=True,
                           timeout=900)
        assert r.returncode == 0, (extra, r.stdout[-1500:], r.stderr[-3000:])
        fps.append(_line(r.stdout)["output_fingerprint"])
    assert fps[0] == fps[1] != 0, fps


def test_bench_two_rank_rehearsal_with_heads_split_by_query_range():
    """--placement split: a full-attention head computes one range of its queries on each of two ranks
    (both receive its K and V); the layer must be the one whole heads give, bit for bit -- 16-bit, e4m3 and int8-score
    (the per-head scales are taken over the whole head on both ranks)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(VORTA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    base = [sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-gemm-ceiling"]
    # wan-tiny: no text tokens; tiny: the Hunyuan form (64 text rows behind the video: the part of a split head that ends at
    # the last video token answers its text queries)
    for config, dtype in (("wan-tiny", "bf16"), ("wan-tiny", "fp8"), ("wan-tiny", "i8pv"), ("tiny", "bf16"), ("tiny", "i8pv")):
        fps = {}
        for placement in ("uneven", "split"):
            r = subprocess.run(base + ["--config", config, "--dtype", dtype, "--placement", placement], cwd=ROOT, env=env,
                               capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, (dtype, placement, r.stdout[-1500:], r.stderr[-3000:])
            j = _line(r.stdout)
            assert j["exchange_selfcheck"]["ok"] is True
            fps[placement] = j["output_fingerprint"]
            if placement == "split":
                par = j["config"]["parallelism"]
                assert "split head placement" in par and "extra parts" in par and "split by query range: 0 extra" not in par, par
                ratio = float(par.split("worst layer: ")[1].split(";")[0])
                assert ratio <= 1.02, par
        assert fps["uneven"] == fps["split"] != 0, (config, dtype, fps)


def test_bench_three_rank_rehearsal_heads_not_divisible():
    """3 ranks sharing the GPU (gloo, host-staged): 8 heads do not divide by 3 -- the reference's reshard and the equal-count
    placement both refuse that -- but head counts that follow the routes place them (3 + 3 + 2 and the like).  One collective
    per tensor with per-rank split sizes, for the whole layer and per slot group, give the same layer, 16-bit and e4m3."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(VORTA_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    base = [sys.executable, "bench.py", "--gpus", "3", "--config", "tiny", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]
    for dtype in ("bf16", "fp8"):
        fps = {}
        for groups in ("1", "2"):  # one collective per tensor with per-rank split sizes, for the layer and per slot group
            r = subprocess.run(base + ["--dtype", dtype, "--sp-groups", groups], cwd=ROOT, env=env,
                               capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, (dtype, groups, r.stdout[-1500:], r.stderr[-3000:])
            j = _line(r.stdout)
            assert j["n_gpus"] == 3 and j["process_group"]["world_size"] == 3 and "uneven" in j["config"]["parallelism"]
            assert j["exchange_selfcheck"]["ok"] is True and "head counts" in j["exchange_selfcheck"]["what"]
            fps[groups] = j["output_fingerprint"]
        assert fps["1"] == fps["2"] != 0, (dtype, fps)
    r = subprocess.run(base + ["--placement", "even"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode != 0  # H % P != 0: the equal-count placement has no answer


def test_bench_emulated_rank_and_fp8_lines():
    """--emulate-rank P: one rank's share of a P-way Ulysses step on one GPU (loopback layout, no transfers);
    --dtype fp8: the e4m3 path (quantiser inside the step)."""
    r = subprocess.run([sys.executable, "bench.py", "--config", "tiny", "--steps", "1", "--warmup", "1", "--emulate-rank", "4",
                        "--no-gemm-ceiling"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r.stdout)
    assert j["emulated_rank_of"] == 4 and j["n_gpus"] == 1 and "emulated" in j["config"]["parallelism"] and "cpu_baseline" not in j
    r = subprocess.run([sys.executable, "bench.py", "--config", "tiny", "--steps", "1", "--warmup", "1", "--dtype", "fp8",
                        "--no-gemm-ceiling", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r.stdout)
    assert j["dtype"] == "fp8" and j["roofline"]["peak"] == 5000.0 and j["roofline"]["kernel"].startswith("attn8_")


def test_bench_emulated_rank_fp8_line():
    r = subprocess.run([sys.executable, "bench.py", "--config", "tiny", "--steps", "1", "--warmup", "1", "--emulate-rank", "2",
                        "--dtype", "fp8", "--no-gemm-ceiling"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r.stdout)
    assert j["emulated_rank_of"] == 2 and j["dtype"] == "fp8" and j["roofline"]["kernel"].startswith("attn8_")


def test_bench_processor_level_line():
    """--level processor: the route plan and every block's processor __call__ inside the timed step, under
    torch.cuda.set_sync_debug_mode("error") -- a host synchronisation anywhere on that path fails the run."""
    for extra in ([], ["--dtype", "fp8"], ["--config", "wan-tiny"]):
        r = subprocess.run([sys.executable, "bench.py", "--config", "tiny", "--steps", "2", "--warmup", "1", "--level", "processor",
                            "--no-gemm-ceiling"] + extra, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (extra, r.stderr[-3000:])
        j = _line(r.stdout)
        assert j["config"]["call_path"]["level"] == "processor" and j["config"]["call_path"]["sync_debug_mode"] == "error"
        assert j["value"] > 0 and "cpu_baseline" not in j and j["roofline"]["achieved"] > 0


def test_bench_multi_gpu_path_end_to_end_on_rccl_with_one_rank():
    """The whole `bench.py --gpus N` path on a REAL RCCL process group -- the launcher, the GPU-free supervisor, the per-attempt
    rendezvous store, init_process_group("nccl", device_id=...), the exchange's collectives (all_to_all_single, all-reduce,
    all-gather), the tagged self-check, the timed step, the exchange breakdown, the teardown -- with ONE rank (VORTA_BENCH_FORCE_SP=1
    + VORTA_SP_FORCE_COLLECTIVES=1): RCCL refuses two ranks on one device, so this is the only end-to-end rehearsal of the
    driver's multi-GPU command a one-GPU box allows."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", VORTA_BENCH_FORCE_SP="1", VORTA_SP_FORCE_COLLECTIVES="1")
    base = ["--gpus", "1", "--config", "tiny", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-gemm-ceiling"]
    fps = {}
    for dtype, extra in (("bf16", []), ("bf16", ["--sp-groups", "2"]), ("i8pv", ["--sp-groups", "2"])):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                            "127.0.0.1", "--master-port", str(port), "bench.py"] + base + ["--dtype", dtype] + extra, cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (dtype, r.stdout[-1500:], r.stderr[-3000:])
        assert [l for l in r.stdout.splitlines() if l.strip()] == [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]  # stdout: the line alone
        j = _line(r.stdout)
        assert j["backend"] == "nccl" and j["process_group"]["backend"] == "nccl" and j["process_group"]["world_size"] == 1
        assert j["exchange_selfcheck"]["ok"] is True and j["exchange"]["exchange_ms_per_layer"] > 0 and "fallback" not in j
        assert "ulysses sp1" in j["config"]["parallelism"] and j["output_fingerprint"] != 0
        fps[(dtype, tuple(extra))] = j["output_fingerprint"]
    assert fps[("bf16", ())] == fps[("bf16", ("--sp-groups", "2"))]  # one slot group and two: the same layer, bit for bit
