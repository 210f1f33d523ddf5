# Round 4: heads split by query range (--placement split): the 2-rank rehearsal test, then the heaviest rank of 8 at Wan-14B-81f
# with whole heads (uneven) and with split heads, 16-bit / e4m3 / int8-score, same box.
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4/split
rm -rf $O && mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_hip_bench.py -x -q -m gpu -k "split_by_query_range or single_gpu_line or two_rank_rehearsal" > $O/pytest.txt 2>&1 || (tail -60 $O/pytest.txt; exit 1)
tail -2 $O/pytest.txt
B="python3 bench.py --config wan14b-81f --emulate-rank 8 --no-gemm-ceiling --steps 2 --warmup 1"
for dt in fp8 i8pv bf16; do
  for pl in uneven split; do
    $B --dtype $dt --placement $pl > $O/rank_of_8_${dt}_$pl.json 2>> $O/err.txt || tail -5 $O/err.txt
    echo "$dt $pl done"
  done
done
python3 bench.py --config wan14b-81f --dtype fp8 --no-gemm-ceiling --no-cpu-baseline --steps 2 --warmup 1 > $O/one_gpu_fp8.json 2>> $O/err.txt
for f in $O/*.json; do python3 -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['dtype'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['share_of_step'], d['config']['parallelism'][-150:])"; done | tee $O/summary.txt
