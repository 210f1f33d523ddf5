# Round 4: first run of the int8-score path: operand-map probe, its tests, then a first bench line next to bf16 / fp8 / fp8pv.
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4/i8
rm -rf "$O" && mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -w $R/tools/probe_i8_layout.hip -o /tmp/probe_i8 && /tmp/probe_i8 | tee $O/probe_i8_layout.txt
cd $R
timeout -k 10 900 python3 -m pytest tests/test_hip_i8.py -x -q -s > $O/pytest_i8.txt 2>&1 || (tail -80 $O/pytest_i8.txt; exit 1)
tail -40 $O/pytest_i8.txt
for d in bf16 fp8 fp8pv i8pv; do
  python3 bench.py --config wan14b-81f --dtype $d --steps 2 --warmup 1 --no-cpu-baseline --no-gemm-ceiling > $O/bench_wan14b_$d.json 2>> $O/bench.err || tail -5 $O/bench.err
done
for f in $O/bench_*.json; do python3 -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"; done | tee $O/summary.txt
