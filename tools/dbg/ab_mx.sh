# A/B of the mixed-precision kernel's build knobs (suffixed libraries built beside the product one), alternating runs:
#   for v in "p3 -DVORTA_MX_PV_VALU=3" ...; do VORTA_BUILD_SUFFIX=_$n VORTA_EXTRA_FLAGS="..." python -m vorta_amd.build; done
set -eux
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3/mxab2
rm -rf $O && mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-gemm-ceiling --steps 2 --warmup 1 --config wan14b-81f --dtype fp8pv"
for rep in 1 2; do
  $B > $O/base_$rep.json 2>> $O/err.txt
  for lib in vorta_amd/csrc/libvorta_hip_*.so; do
    v=$(basename $lib .so | sed 's/libvorta_hip//')
    VORTA_HIP_LIB=$PWD/$lib $B > $O/v${v}_$rep.json 2>> $O/err.txt
  done
done
for f in $O/*.json; do python3 -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['switches']['library'][-50:])"; done
