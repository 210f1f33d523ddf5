# Round 4: the e4m3 kernel with the row-sum MFMA first in the matrix part against last (VARIANTS), same box, alternating runs:
# parity tests of the fp8 path on the first variant, then Wan-14B-81f and Hunyuan-129f fused layers.
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4/fp8ab
rm -rf $O && mkdir -p $O
set -- $VARIANTS
VORTA_HIP_LIB=$PWD/vorta_amd/csrc/libvorta_hip_$1.so timeout -k 10 600 python3 -m pytest tests/test_hip_fp8.py -x -q -m gpu > $O/pytest_fp8_$1.txt 2>&1 || (tail -40 $O/pytest_fp8_$1.txt; exit 1)
tail -2 $O/pytest_fp8_$1.txt
B="python3 bench.py --no-cpu-baseline --no-gemm-ceiling --steps 2 --warmup 1 --dtype fp8"
for rep in 1 2; do
  for v in $VARIANTS; do
    VORTA_HIP_LIB=$PWD/vorta_amd/csrc/libvorta_hip_$v.so $B --config wan14b-81f > $O/${v}_wan_$rep.json 2>> $O/err.txt || tail -3 $O/err.txt
    VORTA_HIP_LIB=$PWD/vorta_amd/csrc/libvorta_hip_$v.so $B --config hunyuan-129f > $O/${v}_hy_$rep.json 2>> $O/err.txt || tail -3 $O/err.txt
  done
done
for f in $O/*.json; do python3 -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['dtype'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['output_fingerprint'], d['switches']['library'][-40:])"; done | tee $O/summary.txt
