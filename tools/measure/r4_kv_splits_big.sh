# Round 4: do key splits also help the ranks of the big configurations (5-6 rounds of workgroups per layer)?  heaviest rank of 8, same box
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4/kvs_big
rm -rf $O && mkdir -p $O
for spec in "wan14b-81f bf16" "wan14b-81f i8pv" "hunyuan-129f fp16"; do
  set -- $spec
  for s in 1 2 3; do
    python3 bench.py --config $1 --dtype $2 --emulate-rank 8 --no-gemm-ceiling --steps 2 --warmup 1 --kv-splits $s > $O/$1_$2_s$s.json 2>> $O/err.txt || tail -5 $O/err.txt
    echo "$spec $s done"
  done
done
for f in $O/*.json; do python3 -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['dtype'], d['ms_per_step'], d['roofline']['share_of_step'])"; done | tee $O/summary.txt
