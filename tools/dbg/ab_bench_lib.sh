# same-box A/B of bench.py lines between the product library and suffixed builds
# usage: VARIANTS="base _m2" CONFIGS="wan14b-81f:fp8 hunyuan-129f:fp8" bash tools/dbg/ab_bench_lib.sh
mkdir -p gpurun_out/r3/abl
B="--steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-gemm-ceiling"
for r in a b; do
for c in ${CONFIGS:-wan14b-81f:fp8}; do
for v in ${VARIANTS:-base}; do
  s=$v; [ "$v" = base ] && s=""
  VORTA_HIP_LIB=vorta_amd/csrc/libvorta_hip$s.so python3 bench.py $B --config ${c%%:*} --dtype ${c##*:} > gpurun_out/r3/abl/${c%%:*}_${c##*:}_${v}_$r.json 2>> gpurun_out/r3/abl/err.txt
  python3 -c "
import json; d=json.loads(open('gpurun_out/r3/abl/${c%%:*}_${c##*:}_${v}_$r.json').read().strip().splitlines()[-1]); print('${c} ${v} ${r}', d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
done; done; done
