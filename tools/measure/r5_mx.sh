# Round 5: the int8-score kernel with MX-scaled probabilities: its parity tests (emulator, routed op, full-size sampled waves),
# the accuracy gate table (PSNR + relative error on every input family), then same-box alternating bench lines against the round-4
# kernel (libvorta_hip_old.so = round 4's attn_fwd_i8.hip beside this tree's other objects).
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5/mx${MX_TAG:-}
rm -rf $O && mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_hip_i8.py -x -q -s -k "not psnr" > $O/pytest_i8.txt 2>&1 || (tail -60 $O/pytest_i8.txt; exit 1)
tail -3 $O/pytest_i8.txt
timeout -k 10 900 python3 -m pytest tests/test_hip_i8.py -q -s -k "psnr" > $O/pytest_i8_psnr.txt 2>&1 || (grep -E "i8pv vs|failures|assert" $O/pytest_i8_psnr.txt | tail -40)
grep "i8pv vs" $O/pytest_i8_psnr.txt > $O/i8pv_psnr_families.txt || true
tail -3 $O/pytest_i8_psnr.txt
timeout -k 10 900 python3 -m pytest tests/test_hip_configs.py -x -q -k "i8" > $O/pytest_configs_i8.txt 2>&1 || (tail -40 $O/pytest_configs_i8.txt; exit 1)
tail -2 $O/pytest_configs_i8.txt
B="python3 bench.py --no-cpu-baseline --no-gemm-ceiling --steps 2 --warmup 1"
for rep in 1 2; do
  for c in wan14b-81f:i8pv hunyuan-129f:i8pv; do
    for v in base old; do
      s=_$v; [ "$v" = base ] && s=""
      VORTA_HIP_LIB=$PWD/vorta_amd/csrc/libvorta_hip$s.so $B --config ${c%%:*} --dtype ${c##*:} > $O/${c%%:*}_${c##*:}_${v}_$rep.json 2>> $O/err.txt || tail -3 $O/err.txt
    done
  done
done
for f in $O/*.json; do python3 -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['dtype'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['output_fingerprint'])"; done | tee $O/summary.txt
