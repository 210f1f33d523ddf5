# Round 5 measurement set on one box (VORTA_TREE_HEAD = git head of the tree, passed in by the caller: the box has no .git):
# the default bench line plain (cpu_baseline, library GEMM and borrowed-SDPA context) and under rocprofv3 --kernel-trace --stats,
# FETCH_SIZE / WRITE_SIZE / MFMA-busy / GUI-active passes (one counter family per pass, no trace domains) for the headline and
# the int8-score workload, the int8-score line under --stats, the same-box precision table at Wan-14B-81f, the heaviest rank of 8.
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun (it exports GRAFT_REPO_ROOT)}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5/round
PART=${PART:-AB}   # A: the default line, --stats, counter passes and their tables; B: the other lines (one gpurun call each: 20 min limit)
[ "$PART" = B ] || rm -rf "$O"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
NB="--no-cpu-baseline --no-gemm-ceiling"
echo "${VORTA_TREE_HEAD:-unknown}" > $O/tree_head.txt
if [ "$PART" != B ]; then
# 1. default bench line, un-profiled
python3 $R/bench.py > $O/bench_hunyuan129f_uniform.json 2> $O/bench_default.err || tail -5 $O/bench_default.err
echo "default line done"
# 2. the same command under rocprofv3 --kernel-trace --stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_hy -- python3 $R/bench.py --steps 1 --warmup 1 $NB > $O/bench_hunyuan129f_uniform_under_rocprof.json 2> $O/stats_hy.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_i8 -- python3 $R/bench.py --config wan14b-81f --dtype i8pv --steps 1 --warmup 1 $NB > $O/bench_wan14b-81f_i8pv_under_rocprof.json 2> $O/stats_i8.err
echo "stats done"
# 3. traffic passes, keyed as tools/pmc_traffic_table.py expects (<config>_<mix>_<dtype>/<COUNTER>)
while read -r cfg mix dt; do
  tag="${cfg}_${mix}_${dt}"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 280 rocprofv3 --pmc $c --output-format csv -d $O/pmc/$tag/$c -- python3 $R/bench.py --config $cfg --mix $mix --dtype $dt --steps 1 --warmup 0 $NB > $O/pmc_$tag.$c.json 2> $O/pmc_$tag.$c.err
    echo "done $tag $c"
  done
  timeout -k 10 280 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $O/pmc_mfma_$tag -- python3 $R/bench.py --config $cfg --mix $mix --dtype $dt --steps 1 --warmup 0 $NB > /dev/null 2> $O/pmc_mfma_$tag.err
  timeout -k 10 280 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_gui_$tag -- python3 $R/bench.py --config $cfg --mix $mix --dtype $dt --steps 1 --warmup 0 $NB > /dev/null 2> $O/pmc_gui_$tag.err
  echo "done $tag mfma/gui"
done <<'LIST'
hunyuan-129f uniform fp16
wan14b-81f uniform i8pv
LIST
cd $R
python3 tools/pmc_traffic_table.py $O/pmc --json $O/r05_pmc_traffic.json --head "$(cat $O/tree_head.txt)" --source tools/measure/r5_round.sh > $O/pmc_traffic_table.txt
python3 tools/pmc_summary.py $O/pmc/hunyuan-129f_uniform_fp16/FETCH_SIZE $O/pmc/hunyuan-129f_uniform_fp16/WRITE_SIZE $O/pmc_mfma_hunyuan-129f_uniform_fp16 $O/pmc_gui_hunyuan-129f_uniform_fp16 --match attn --json $O/pmc_hunyuan_fp16.json > $O/pmc_hunyuan_fp16.txt || true
python3 tools/pmc_summary.py $O/pmc/wan14b-81f_uniform_i8pv/FETCH_SIZE $O/pmc/wan14b-81f_uniform_i8pv/WRITE_SIZE $O/pmc_mfma_wan14b-81f_uniform_i8pv $O/pmc_gui_wan14b-81f_uniform_i8pv --match attn_i8 --json $O/pmc_wan14b_i8pv.json > $O/pmc_wan14b_i8pv.txt || true
find $O -name "*counter_collection.csv" -size +8M -delete || true
cd /tmp
fi
if [ "$PART" != A ]; then
# 4. same-box precision table at Wan-14B-81f and the other configurations' lines
for c in "wan14b-81f bf16" "wan14b-81f fp8pv" "wan14b-81f i8pv" "wan14b-81f fp8" "wan1.3b-81f bf16" "hunyuan-129f bf16" "hunyuan-129f i8pv" "hunyuan-117f bf16"; do
  set -- $c; python3 $R/bench.py --config $1 --dtype $2 --steps 2 --warmup 1 $NB > $O/bench_$1_$2.json 2>> $O/bench_cfg.err; echo "line $c done"; done
# 5. the heaviest rank of 8 (one GPU, no transfers) and the processor-level line
python3 $R/bench.py --config wan14b-81f --dtype i8pv --emulate-rank 8 --no-gemm-ceiling --steps 2 --warmup 1 > $O/rank_of_8_wan14b_i8pv.json 2>> $O/bench_cfg.err || true
python3 $R/bench.py --config wan14b-81f --dtype bf16 --emulate-rank 8 --no-gemm-ceiling --steps 2 --warmup 1 > $O/rank_of_8_wan14b_bf16.json 2>> $O/bench_cfg.err || true
python3 $R/bench.py --emulate-rank 8 --no-gemm-ceiling --steps 2 --warmup 1 > $O/rank_of_8_hunyuan_fp16.json 2>> $O/bench_cfg.err || true
python3 $R/bench.py --level processor --no-gemm-ceiling --steps 2 --warmup 1 > $O/processor_hunyuan_fp16.json 2>> $O/bench_cfg.err || true
fi
cd $R
for f in $O/bench_*.json $O/rank_*.json $O/processor_*.json; do python3 -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; print('$f'.split('/')[-1], d['dtype'], d['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac'], r['share_of_step'], r.get('library_sdpa_tflops'), r.get('own_dense_tflops_same_sample'))"; done | tee $O/summary.txt
# keep the merged output small: the raw counter csvs are large
find $O -name "*counter_collection.csv" -size +8M -delete || true
du -sh $O
