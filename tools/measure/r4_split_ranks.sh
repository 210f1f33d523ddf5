# Round 4: --placement split against whole heads at 3 and 4 ranks sharing one GPU (gloo, host-staged): one fingerprint per precision
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4/split_ranks
rm -rf $O && mkdir -p $O
export VORTA_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 3 4; do
  for dt in bf16 i8pv fp8; do
    for pl in uneven split; do
      for g in 1 2; do
        timeout -k 10 300 python3 bench.py --gpus $n --config wan-tiny --dtype $dt --placement $pl --sp-groups $g --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-ceiling > $O/n${n}_${dt}_${pl}_g$g.json 2>> $O/err.txt || (tail -20 $O/err.txt; exit 1)
      done
    done
  done
done
for f in $O/*.json; do python3 -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['output_fingerprint'], d['exchange_selfcheck']['ok'], d['config']['parallelism'][-120:])"; done | tee $O/summary.txt
