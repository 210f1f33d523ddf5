# Round 5: the whole GPU suite of the working tree, then the default bench line (with the borrowed-SDPA context) and the
# int8-score / e4m3 lines of Wan-14B-81f on the same box (the A/B base of the round's 8-bit loop work)
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5/suite${SUITE_TAG:-}
rm -rf $O && mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -q -m gpu -x > $O/pytest_gpu.txt 2>&1 || (tail -80 $O/pytest_gpu.txt; exit 1)
tail -3 $O/pytest_gpu.txt
python3 bench.py > $O/bench_default.json 2>> $O/err.txt || tail -5 $O/err.txt
B="--no-cpu-baseline --steps 2 --warmup 1"
python3 bench.py --config wan14b-81f --dtype i8pv $B > $O/bench_wan14b_i8pv.json 2>> $O/err.txt || tail -5 $O/err.txt
python3 bench.py --config wan14b-81f --dtype fp8 $B > $O/bench_wan14b_fp8.json 2>> $O/err.txt || tail -5 $O/err.txt
python3 bench.py --config wan14b-81f --dtype bf16 $B > $O/bench_wan14b_bf16.json 2>> $O/err.txt || tail -5 $O/err.txt
for f in $O/*.json; do python3 -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; print('$f', d['dtype'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac'], r['share_of_step'], r.get('library_gemm_tflops'), r.get('library_sdpa_tflops'), r.get('own_dense_tflops_same_sample'), d['output_fingerprint'])"; done | tee $O/summary.txt
