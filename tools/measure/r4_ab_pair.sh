# Round 4: paired steps (ring 4: one burst of tile requests and one vmcnt(0) + barrier per TWO key blocks) against the product
# loop, same box, alternating runs of the fused fp16 layer; first the parity tests of the 16-bit kernels on the variant.
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4/ab_pair
rm -rf $O && mkdir -p $O
V=${VARIANTS:-_r4}
for v in $V; do
  VORTA_HIP_LIB=$PWD/vorta_amd/csrc/libvorta_hip$v.so timeout -k 10 600 python3 -m pytest tests/test_hip_attention.py tests/test_hip_experts.py tests/test_hip_configs.py -x -q -m gpu > $O/pytest$v.txt 2>&1 || (tail -60 $O/pytest$v.txt; exit 1)
  tail -2 $O/pytest$v.txt
done
B="python3 bench.py --no-cpu-baseline --no-gemm-ceiling --steps 2 --warmup 1"
for rep in 1 2 3; do
  for v in base $V; do
    s=$v; [ "$v" = base ] && s=""
    VORTA_HIP_LIB=$PWD/vorta_amd/csrc/libvorta_hip$s.so $B > $O/${v}_fp16_$rep.json 2>> $O/err.txt
    VORTA_HIP_LIB=$PWD/vorta_amd/csrc/libvorta_hip$s.so $B --config wan14b-81f --dtype bf16 > $O/${v}_wan_bf16_$rep.json 2>> $O/err.txt
  done
done
for f in $O/*.json; do python3 -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['output_fingerprint'] if 'output_fingerprint' in d else '', d['switches']['library'][-40:])"; done | tee $O/summary.txt
