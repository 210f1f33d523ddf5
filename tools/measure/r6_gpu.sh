# Round 6: one parameterised script per kind of GPU call (PART=...), run through gpurun from the repo root:
#   PART=tests   the GPU suite (verbose: a line per test), with a heartbeat so a long quiet test is not taken for a hang
#   PART=probe   tools/probe_cvt_pknorm.hip (what v_cvt_pknorm_u16_f32 rounds to and costs)
#   PART=shape   tools/probe_mfma_shape.hip part F (SHAPE_PARTS=32): the 16x16x32 step at 64 query rows per wave
#   PART=i8ab    same-box alternating A/B of int8-score kernel variants (VARIANTS="pkn ...": libvorta_hip_<name>.so)
#   PART=bench   the round's bench lines (+ rocprofv3 kernel stats of the default command)
#   PART=psnr    tools/structured_psnr.py tables
#   PART=pmc     counter passes (FETCH_SIZE, WRITE_SIZE, MFMA-busy family, GRBM_GUI_ACTIVE: one family per pass, no trace domains)
#                of WORKLOADS="cfg:mix:dtype ..." and the tables tools/pmc_traffic_table.py / pmc_summary.py fold them into
#   PART=fullsp  BASELINE configs[3] / [4] at FULL SIZE through the N > 1 code path on ONE GPU: RANKS (default 2) ranks sharing the GPU,
#                gloo host-staged messages (VORTA_BENCH_BACKEND=gloo): the tagged self-check, the step and the exchange breakdown on the
#                production shapes (the rates say nothing about xGMI; the bytes and the index maps are the production's)
#   PART=lines   the other configurations' lines + the heaviest rank of 8 + the processor-level line (LINES="cfg:dtype ...")
# VORTA_TREE_HEAD = git head of the tree (the box has no .git), stamped into the traffic table.
set -ux
cd "${GRAFT_REPO_ROOT:?run through gpurun}"
O=gpurun_out/r6
mkdir -p $O
( while true; do date +%s >> $O/heartbeat.txt; sleep 60; done ) &
HB=$!
trap "kill $HB" EXIT
case "${PART:-tests}" in
tests)
  timeout -k 10 ${LIMIT:-1000} python3 -m pytest ${TESTS:-tests} -m gpu -v -x --durations=15 > $O/pytest_${TAG:-all}.txt 2>&1
  echo "rc=$?" >> $O/pytest_${TAG:-all}.txt
  tail -25 $O/pytest_${TAG:-all}.txt
  ;;
probe)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probe_cvt_pknorm.hip -o /tmp/probe_cvt_pknorm 2> /dev/null
  timeout -k 10 120 /tmp/probe_cvt_pknorm | tee $O/probe_cvt_pknorm.txt
  ;;
shape)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -mllvm -enable-post-misched=0 tools/probe_mfma_shape.hip -o /tmp/probe_shape 2> /dev/null
  timeout -k 10 600 /tmp/probe_shape ${SHAPE_PARTS:-32} 2 | tee $O/probe_mfma_shape_partF.txt
  ;;
i8ab)
  set -- $VARIANTS
  for v in ${TEST_VARIANTS:-$1}; do
    VORTA_HIP_LIB=$PWD/vorta_amd/csrc/libvorta_hip_$v.so timeout -k 10 600 python3 -m pytest tests/test_hip_i8.py -x -q > $O/pytest_i8_$v.txt 2>&1 || { tail -40 $O/pytest_i8_$v.txt; exit 1; }
    tail -2 $O/pytest_i8_$v.txt
  done
  B="python3 bench.py --no-cpu-baseline --no-gemm-ceiling --steps ${STEPS:-2} --warmup 1"
  for rep in 1 2 3; do
    for c in ${CONFIGS:-wan14b-81f:i8pv}; do
      for v in base $VARIANTS; do
        s=_$v; [ "$v" = base ] && s=""
        VORTA_HIP_LIB=$PWD/vorta_amd/csrc/libvorta_hip$s.so timeout -k 10 300 $B --config ${c%%:*} --dtype ${c##*:} > $O/ab_${c%%:*}_${c##*:}_${v}_$rep.json 2>> $O/ab_err.txt || tail -3 $O/ab_err.txt
      done
    done
  done
  for f in $O/ab_*.json; do python3 -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['dtype'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['output_fingerprint'])"; done | tee $O/ab_summary.txt
  ;;
bench)
  for c in ${CONFIGS:-hunyuan-129f:fp16 wan1.3b-49f:bf16}; do
    timeout -k 10 600 python3 bench.py --config ${c%%:*} --dtype ${c##*:} --steps ${STEPS:-5} --warmup 2 > $O/bench_${c%%:*}_${c##*:}.json 2>> $O/bench_err.txt || tail -5 $O/bench_err.txt
    tail -c 600 $O/bench_${c%%:*}_${c##*:}.json
  done
  if [ -n "${PROF:-}" ]; then
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gemm-ceiling > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof_err.txt
    cd $GRAFT_REPO_ROOT
    find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
    head -5 $O/bench_kernel_stats.csv
  fi
  ;;
psnr)
  for c in ${CONFIGS:-hunyuan-129f wan14b-81f}; do
    timeout -k 10 900 python3 tools/structured_psnr.py --config $c --heads ${HEADS:-12} --precision ${PRECISION:-native} > $O/structured_psnr_${c}_${PRECISION:-native}.txt 2>> $O/psnr_err.txt || tail -5 $O/psnr_err.txt
    tail -30 $O/structured_psnr_${c}_${PRECISION:-native}.txt
  done
  ;;
pmc)
  R=$PWD; OO=$R/$O/pmc_run; mkdir -p $OO
  cd /tmp && export TMPDIR=/tmp
  NB="--no-cpu-baseline --no-gemm-ceiling"
  for w in ${WORKLOADS:-hunyuan-129f:uniform:fp16 wan14b-81f:uniform:i8pv}; do
    IFS=: read -r cfg mix dt <<< "$w"
    tag="${cfg}_${mix}_${dt}"
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout -k 10 280 rocprofv3 --pmc $c --output-format csv -d $OO/pmc/$tag/$c -- python3 $R/bench.py --config $cfg --mix $mix --dtype $dt --steps 1 --warmup 0 $NB > $OO/pmc_$tag.$c.json 2> $OO/pmc_$tag.$c.err
      echo "done $tag $c"
    done
    timeout -k 10 280 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OO/pmc_mfma_$tag -- python3 $R/bench.py --config $cfg --mix $mix --dtype $dt --steps 1 --warmup 0 $NB > /dev/null 2> $OO/pmc_mfma_$tag.err
    timeout -k 10 280 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OO/pmc_gui_$tag -- python3 $R/bench.py --config $cfg --mix $mix --dtype $dt --steps 1 --warmup 0 $NB > /dev/null 2> $OO/pmc_gui_$tag.err
    echo "done $tag mfma/gui"
  done
  cd $R
  python3 tools/pmc_traffic_table.py $OO/pmc --json $O/r06_pmc_traffic.json --head "${VORTA_TREE_HEAD:-unknown}" --source tools/measure/r6_gpu.sh > $O/r06_pmc_traffic_table.txt
  for w in ${WORKLOADS:-hunyuan-129f:uniform:fp16 wan14b-81f:uniform:i8pv}; do
    IFS=: read -r cfg mix dt <<< "$w"
    tag="${cfg}_${mix}_${dt}"
    python3 tools/pmc_summary.py $OO/pmc/$tag/FETCH_SIZE $OO/pmc/$tag/WRITE_SIZE $OO/pmc_mfma_$tag $OO/pmc_gui_$tag --match attn --json $O/r06_pmc_$tag.json > $O/r06_pmc_$tag.txt || true
    tail -12 $O/r06_pmc_$tag.txt
  done
  rm -rf $OO/pmc $OO/pmc_mfma_* $OO/pmc_gui_*  # the raw counter csvs are large
  ;;
fullsp)
  for c in ${CONFIGS:-hunyuan-129f:fp16}; do
    VORTA_BENCH_BACKEND=gloo VORTA_BENCH_BREAKDOWN_REPS=1 VORTA_BENCH_TIMEOUT_S=600 timeout -k 10 ${LIMIT:-1000} python3 bench.py --gpus ${RANKS:-2} \
      --config ${c%%:*} --dtype ${c##*:} --steps 1 --warmup 0 --no-cpu-baseline ${EXTRA:-} > $O/fullsp_${c%%:*}_${c##*:}_p${RANKS:-2}.json 2> $O/fullsp_err.txt
    echo "rc=$?"; tail -3 $O/fullsp_err.txt | cut -c1-300
    python3 -c "
import json; d=json.loads([l for l in open('$O/fullsp_${c%%:*}_${c##*:}_p${RANKS:-2}.json') if l.startswith('{')][-1]); print({k: d.get(k) for k in ('n_gpus','ms_per_step','exchange_selfcheck','exchange','fallback','error')})"
  done
  ;;
lines)
  NB="--no-cpu-baseline --no-gemm-ceiling"
  for c in ${LINES:-wan14b-81f:bf16 wan14b-81f:fp8pv wan14b-81f:i8pv wan14b-81f:auto8 wan14b-81f:fp8 wan1.3b-81f:bf16 hunyuan-129f:bf16 hunyuan-129f:i8pv}; do
    timeout -k 10 300 python3 bench.py --config ${c%%:*} --dtype ${c##*:} --steps 2 --warmup 1 $NB > $O/line_${c%%:*}_${c##*:}.json 2>> $O/lines_err.txt; echo "line $c done"
  done
  timeout -k 10 300 python3 bench.py --config wan14b-81f --dtype i8pv --emulate-rank 8 --no-gemm-ceiling --steps 2 --warmup 1 > $O/line_rank_of_8_wan14b_i8pv.json 2>> $O/lines_err.txt || true
  timeout -k 10 300 python3 bench.py --emulate-rank 8 --no-gemm-ceiling --steps 2 --warmup 1 > $O/line_rank_of_8_hunyuan_fp16.json 2>> $O/lines_err.txt || true
  timeout -k 10 300 python3 bench.py --level processor --no-gemm-ceiling --steps 2 --warmup 1 > $O/line_processor_hunyuan_fp16.json 2>> $O/lines_err.txt || true
  for f in $O/line_*.json; do python3 -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; print('$f'.split('/')[-1], d['dtype'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac'], r['share_of_step'])"; done | tee $O/lines_summary.txt
  ;;
esac
