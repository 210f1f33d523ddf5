# Round 4: same-box alternating A/B of library variants on bench.py lines.  VARIANTS="a b" (libvorta_hip_<v>.so), ARGS="bench args"
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4/ab_${TAG:-generic}
rm -rf $O && mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-gemm-ceiling --steps 2 --warmup 1 ${ARGS:-}"
for rep in 1 2 3; do
  for v in $VARIANTS; do
    VORTA_HIP_LIB=$PWD/vorta_amd/csrc/libvorta_hip_$v.so $B > $O/${v}_$rep.json 2>> $O/err.txt || tail -3 $O/err.txt
  done
done
for f in $O/*.json; do python3 -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['dtype'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['output_fingerprint'], d['switches']['library'][-40:])"; done | tee $O/summary.txt
