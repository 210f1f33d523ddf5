# Round 4: which half of the workgroup starts with the VALU part in the int8-score kernel (VORTA_I8_SWAP), Wan-14B-81f and Hunyuan-129f
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4/i8swap
rm -rf $O && mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-gemm-ceiling --steps 2 --warmup 1 --dtype i8pv"
for rep in 1 2; do
  for v in sw0 sw1; do
    VORTA_HIP_LIB=$PWD/vorta_amd/csrc/libvorta_hip_$v.so $B --config wan14b-81f > $O/${v}_wan_$rep.json 2>> $O/err.txt || tail -3 $O/err.txt
    VORTA_HIP_LIB=$PWD/vorta_amd/csrc/libvorta_hip_$v.so $B --config hunyuan-129f > $O/${v}_hy_$rep.json 2>> $O/err.txt || tail -3 $O/err.txt
  done
done
for f in $O/*.json; do python3 -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['dtype'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['output_fingerprint'])"; done | tee $O/summary.txt
