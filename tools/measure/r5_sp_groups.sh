# Round 5, VERDICT r04 item 7: rank 1 / 2 / 3 slot groups of the sequence-parallel exchange per config on ONE GPU: the heaviest
# rank of 8 with an EMULATED wire (VORTA_SP_EMULATE_LINK_GBPS: every collective holds a side stream for the time its largest
# chunk needs at that rate per xGMI link + 10 us; vorta_amd/ulysses/engine.py).  An assumption about the links, not a measurement.
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5/spgroups
rm -rf $O && mkdir -p $O
B="--emulate-rank 8 --no-gemm-ceiling --no-cpu-baseline --steps 2 --warmup 1"
for c in "hunyuan-129f fp16" "wan14b-81f i8pv" "wan14b-81f bf16"; do
  set -- $c
  for link in 0 60 150; do
    for g in 1 2 3; do
      [ "$link" = 0 ] && [ "$g" != 1 ] && continue
      VORTA_SP_EMULATE_LINK_GBPS=$link python3 bench.py --config $1 --dtype $2 --sp-groups $g $B > $O/$1_$2_link${link}_g$g.json 2>> $O/err.txt || tail -3 $O/err.txt
    done
  done
done
for f in $O/*.json; do python3 -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['dtype'], d['ms_per_step'], d['output_fingerprint'])"; done | tee $O/summary.txt
