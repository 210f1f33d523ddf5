# Round 4: --placement split where whole heads are a coarse unit: Wan-1.3B-81f (12 heads) on 8 ranks, heaviest rank, same box
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4/split_small
rm -rf $O && mkdir -p $O
B="python3 bench.py --config wan1.3b-81f --emulate-rank 8 --no-gemm-ceiling --steps 3 --warmup 1 ${EXTRA:-}"
for dt in bf16 i8pv; do
  for pl in uneven split; do
    for rep in 1 2; do
      $B --dtype $dt --placement $pl > $O/${dt}_${pl}_$rep.json 2>> $O/err.txt || tail -5 $O/err.txt
    done
  done
done
python3 bench.py --config wan1.3b-81f --no-gemm-ceiling --no-cpu-baseline --steps 3 --warmup 1 > $O/one_gpu_bf16.json 2>> $O/err.txt
for f in $O/*.json; do python3 -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['dtype'], d['ms_per_step'], d['roofline']['share_of_step'], d['config']['parallelism'][-130:])"; done | tee $O/summary.txt
