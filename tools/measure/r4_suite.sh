# Round 4: the whole GPU suite of the working tree, then the i8pv lines (rank of 8, processor level) and the default bench line
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4/suite
rm -rf $O && mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1 || (tail -80 $O/pytest_gpu.txt; exit 1)
tail -3 $O/pytest_gpu.txt
B="--no-gemm-ceiling --steps 2 --warmup 1"
python3 bench.py --config wan14b-81f --dtype i8pv --emulate-rank 8 $B > $O/rank_of_8_wan14b_i8pv.json 2>> $O/err.txt || tail -5 $O/err.txt
python3 bench.py --config wan14b-81f --dtype bf16 --emulate-rank 8 $B > $O/rank_of_8_wan14b_bf16.json 2>> $O/err.txt || tail -5 $O/err.txt
python3 bench.py --config wan14b-81f --dtype i8pv --level processor $B > $O/processor_wan14b_i8pv.json 2>> $O/err.txt || tail -5 $O/err.txt
python3 bench.py --level processor $B > $O/processor_hunyuan_fp16.json 2>> $O/err.txt || tail -5 $O/err.txt
python3 bench.py --config hunyuan-129f --dtype i8pv $B --no-cpu-baseline > $O/bench_hunyuan_i8pv.json 2>> $O/err.txt || tail -5 $O/err.txt
python3 bench.py --config hunyuan-129f --dtype bf16 $B --no-cpu-baseline > $O/bench_hunyuan_bf16.json 2>> $O/err.txt || tail -5 $O/err.txt
for f in $O/*.json; do python3 -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['dtype'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['share_of_step'])"; done | tee $O/summary.txt
