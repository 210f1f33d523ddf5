# Round 5 (final tree): the mixed-precision ("fp8pv") Wan-14B-81f step under rocprofv3 --kernel-trace --stats, then MFMA-busy and
# GUI-active counter passes (separate runs, no trace domains) of its fused layer kernel
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun}"
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5/fp8pv_prof; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
NB="--no-cpu-baseline --no-gemm-ceiling"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --config wan14b-81f --dtype fp8pv --steps 1 --warmup 1 $NB > $O/bench_wan14b-81f_fp8pv_under_rocprof.json 2> $O/stats.err
timeout -k 10 280 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $O/mfma -- python3 $R/bench.py --config wan14b-81f --dtype fp8pv --steps 1 --warmup 0 $NB > /dev/null 2> $O/mfma.err
timeout -k 10 280 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/gui -- python3 $R/bench.py --config wan14b-81f --dtype fp8pv --steps 1 --warmup 0 $NB > /dev/null 2> $O/gui.err
cd $R
python3 tools/pmc_summary.py $O/mfma $O/gui --match attn_mx --json $O/pmc_wan14b_fp8pv.json > $O/pmc_wan14b_fp8pv.txt || true
find $O -name "*counter_collection.csv" -size +8M -delete || true
find $O -name "*kernel_stats.csv" | head -2
python3 -c "
import json; d=json.load(open('$O/pmc_wan14b_fp8pv.json'))
for k,v in d.items(): print(k[:50], v['derived'])"
