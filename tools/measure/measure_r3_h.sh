# Round 3, last batch: smoke, MFMA-busy / clock of the mixed-precision kernel, heaviest rank of 8 with precision fp8pv.
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun (it exports GRAFT_REPO_ROOT)}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3/h
rm -rf "$O" && mkdir -p "$O"
cd $R && python3 __graft_entry__.py smoke > $O/smoke.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_mfma -- python3 $R/bench.py --config wan14b-81f --dtype fp8pv --steps 1 --warmup 0 --no-cpu-baseline --no-gemm-ceiling > /dev/null 2> $O/pmc_mfma.err
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_gui -- python3 $R/bench.py --config wan14b-81f --dtype fp8pv --steps 1 --warmup 0 --no-cpu-baseline --no-gemm-ceiling > /dev/null 2> $O/pmc_gui.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --config wan14b-81f --dtype fp8pv --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-ceiling > $O/bench_wan14b_fp8pv_under_rocprof.json 2> $O/stats.err
cd $R
python3 bench.py --config wan14b-81f --dtype fp8pv --emulate-rank 8 --steps 4 --warmup 1 --no-gemm-ceiling > $O/rank_of_8_wan14b_fp8pv.json 2>> $O/err.txt
python3 bench.py --config wan14b-81f --dtype bf16 --emulate-rank 8 --steps 4 --warmup 1 --no-gemm-ceiling > $O/rank_of_8_wan14b_bf16.json 2>> $O/err.txt
python3 bench.py --config wan14b-81f --dtype fp8pv --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_wan14b_fp8pv.json 2>> $O/err.txt
python3 tools/pmc_summary.py $O/pmc_mfma $O/pmc_gui --match attn_mx_multi --json $O/pmc_mx.json | grep -A3 derived | head -8
cat $O/smoke.txt | tail -4
for n in rank_of_8_wan14b_fp8pv rank_of_8_wan14b_bf16 bench_wan14b_fp8pv bench_wan14b_fp8pv_under_rocprof; do python3 -c "
import json; d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['share_of_step'])"; done
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs head -3 | cut -c1-200
