# Round 4: bench lines of the int8-score path next to bf16 / fp8 / fp8pv (Wan-14B-81f, one box), then the processor tests
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4/i8b
rm -rf "$O" && mkdir -p "$O"
cd $R
for d in bf16 i8pv fp8 fp8pv i8pv bf16; do
  python3 bench.py --config wan14b-81f --dtype $d --steps 2 --warmup 1 --no-cpu-baseline --no-gemm-ceiling > $O/bench_wan14b_${d}_$RANDOM.json 2>> $O/bench.err || tail -5 $O/bench.err
done
for f in $O/bench_*.json; do python3 -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['share_of_step'])"; done | tee $O/summary.txt
timeout -k 10 900 python3 -m pytest tests/test_hip_processors.py tests/test_hip_torch_ops.py tests/test_hip_patch.py -x -q > $O/pytest_proc.txt 2>&1 || (tail -80 $O/pytest_proc.txt; exit 1)
tail -3 $O/pytest_proc.txt
