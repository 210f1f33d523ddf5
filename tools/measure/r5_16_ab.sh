# Round 5: variants of the 16-bit kernel (attn_fwd.hip; VARIANTS = suffixes of libvorta_hip_<name>.so) against the product library:
# parity tests on the first variant, then same-box alternating runs of the fused Hunyuan-129f (fp16) / Wan-14B-81f (bf16) layers.
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5/ab16${AB_TAG:-}
rm -rf $O && mkdir -p $O
set -- $VARIANTS
VORTA_HIP_LIB=$PWD/vorta_amd/csrc/libvorta_hip_$1.so timeout -k 10 900 python3 -m pytest ${TESTS:-tests/test_hip_attention.py tests/test_hip_experts.py} -x -q > $O/pytest_$1.txt 2>&1 || (tail -40 $O/pytest_$1.txt; exit 1)
tail -2 $O/pytest_$1.txt
B="python3 bench.py --no-cpu-baseline --no-gemm-ceiling --steps ${STEPS:-2} --warmup 1"
for rep in 1 2; do
  for c in ${CONFIGS:-hunyuan-129f:fp16 wan14b-81f:bf16}; do
    for v in base $VARIANTS; do
      s=_$v; [ "$v" = base ] && s=""
      VORTA_HIP_LIB=$PWD/vorta_amd/csrc/libvorta_hip$s.so $B --config ${c%%:*} --dtype ${c##*:} > $O/${c%%:*}_${c##*:}_${v}_$rep.json 2>> $O/err.txt || tail -3 $O/err.txt
    done
  done
done
for f in $O/*.json; do python3 -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['dtype'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['output_fingerprint'])"; done | tee $O/summary.txt
