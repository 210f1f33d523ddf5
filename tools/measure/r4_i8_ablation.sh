# Round 4: timing ablations of the int8-score step (each variant removes one ingredient and gives WRONG results): where do the
# cycles beside the MFMAs go?  Same box, Wan-14B-81f fused layer, two rounds.
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4/i8abl
rm -rf $O && mkdir -p $O
B="python3 bench.py --config wan14b-81f --dtype i8pv --no-cpu-baseline --no-gemm-ceiling --no-selfcheck --steps 1 --warmup 1"
for rep in 1 2; do
  for v in base ${VARIANTS}; do
    s=_$v; [ "$v" = base ] && s=""
    VORTA_HIP_LIB=$PWD/vorta_amd/csrc/libvorta_hip$s.so timeout -k 10 120 $B > $O/${v}_$rep.json 2>> $O/err.txt || tail -3 $O/err.txt
    echo "$v $rep done"
  done
done
for f in $O/*.json; do python3 -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['switches']['library'][-60:])"; done | tee $O/summary.txt
