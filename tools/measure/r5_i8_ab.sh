# Round 5: variants of the int8-score step (VARIANTS = suffixes of libvorta_hip_<name>.so built by tools/dbg/build_i8_variants.sh)
# against the product library: parity tests on every variant named in TEST_VARIANTS (default: the first), then same-box
# alternating runs of the fused Wan-14B-81f layer (CONFIGS="cfg:dtype ..." for other lines).
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5/i8ab${AB_TAG:-}
rm -rf $O && mkdir -p $O
set -- $VARIANTS
for v in ${TEST_VARIANTS:-$1}; do
  VORTA_HIP_LIB=$PWD/vorta_amd/csrc/libvorta_hip_$v.so timeout -k 10 600 python3 -m pytest ${TESTS:-tests/test_hip_i8.py} -x -q > $O/pytest_$v.txt 2>&1 || (tail -40 $O/pytest_$v.txt; exit 1)
  tail -2 $O/pytest_$v.txt
done
B="python3 bench.py --no-cpu-baseline --no-gemm-ceiling --steps ${STEPS:-2} --warmup 1"
for rep in 1 2; do
  for c in ${CONFIGS:-wan14b-81f:i8pv}; do
    for v in base $VARIANTS; do
      s=_$v; [ "$v" = base ] && s=""
      VORTA_HIP_LIB=$PWD/vorta_amd/csrc/libvorta_hip$s.so $B --config ${c%%:*} --dtype ${c##*:} > $O/${c%%:*}_${c##*:}_${v}_$rep.json 2>> $O/err.txt || tail -3 $O/err.txt
    done
  done
done
for f in $O/*.json; do python3 -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['dtype'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['output_fingerprint'], d['switches']['library'].split('7b800a19466229b8479a78de19143dc33c3ab9b5)')[-1])"; done | tee $O/summary.txt
