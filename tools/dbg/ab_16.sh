# A/B of a suffixed build of the 16-bit kernel against the product library, alternating runs (VARIANT = suffix, default _pk)
set -eux
cd $GRAFT_REPO_ROOT
V=${VARIANT:-_pk}
O=gpurun_out/r3/ab16
rm -rf $O && mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-gemm-ceiling --steps 2 --warmup 1"
for rep in 1 2 3; do
  $B > $O/base_fp16_$rep.json 2>> $O/err.txt
  VORTA_HIP_LIB=$PWD/vorta_amd/csrc/libvorta_hip$V.so $B > $O/var_fp16_$rep.json 2>> $O/err.txt
done
for rep in 1 2; do
  $B --dtype bf16 > $O/base_bf16_$rep.json 2>> $O/err.txt
  VORTA_HIP_LIB=$PWD/vorta_amd/csrc/libvorta_hip$V.so $B --dtype bf16 > $O/var_bf16_$rep.json 2>> $O/err.txt
done
for f in $O/*.json; do python3 -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['switches']['library'][-30:])"; done
