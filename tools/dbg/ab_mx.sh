# A/B of the mixed-precision kernel's build knobs (suffixed libraries built beside the product one), alternating runs
set -eux
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3/mxab
rm -rf $O && mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-gemm-ceiling --steps 2 --warmup 1 --config wan14b-81f --dtype fp8pv"
for rep in 1 2; do
  for v in "" _s0 _k0 _k2; do
    if [ -z "$v" ]; then $B > $O/base_$rep.json 2>> $O/err.txt; else VORTA_HIP_LIB=$PWD/vorta_amd/csrc/libvorta_hip$v.so $B > $O/v${v}_$rep.json 2>> $O/err.txt; fi
  done
done
for f in $O/*.json; do python3 -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['switches']['library'][-40:])"; done
