# Round 4: the int8-score step with every score-phase read issued at the top (variants v3*) against the committed step:
# parity tests on the first variant, then same-box alternating runs of the fused Wan-14B-81f layer.
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4/i8ab3
rm -rf $O && mkdir -p $O
set -- $VARIANTS
VORTA_HIP_LIB=$PWD/vorta_amd/csrc/libvorta_hip_$1.so timeout -k 10 600 python3 -m pytest tests/test_hip_i8.py -x -q > $O/pytest_i8_$1.txt 2>&1 || (tail -40 $O/pytest_i8_$1.txt; exit 1)
tail -2 $O/pytest_i8_$1.txt
B="python3 bench.py --config wan14b-81f --no-cpu-baseline --no-gemm-ceiling --steps 2 --warmup 1"
$B --dtype bf16 > $O/bf16_0.json 2>> $O/err.txt
$B --dtype fp8 > $O/fp8_0.json 2>> $O/err.txt
for rep in 1 2; do
  for v in base $VARIANTS; do
    s=_$v; [ "$v" = base ] && s=""
    VORTA_HIP_LIB=$PWD/vorta_amd/csrc/libvorta_hip$s.so $B --dtype i8pv > $O/${v}_$rep.json 2>> $O/err.txt || tail -3 $O/err.txt
  done
done
for f in $O/*.json; do python3 -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['dtype'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['output_fingerprint'], d['switches']['library'][-50:])"; done | tee $O/summary.txt
