set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun (it exports GRAFT_REPO_ROOT)}"
mkdir -p gpurun_out/r2/m
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2/m
cd /tmp && export TMPDIR=/tmp
# 1. default bench, un-profiled (with cpu_baseline and the GEMM context)
python3 $R/bench.py > $O/bench_hunyuan129f_uniform.json 2> $O/bench_default.err || tail -5 $O/bench_default.err
# 2. same command under rocprofv3 --kernel-trace --stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_hy -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-ceiling > $O/bench_hunyuan129f_uniform_under_rocprof.json 2> $O/stats_hy.err
# 3. PMC passes (one family per pass, no trace domains besides kernel-trace)
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-gemm-ceiling > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-gemm-ceiling > $O/pmc_write.json 2> $O/pmc_write.err
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_mfma -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-gemm-ceiling > $O/pmc_mfma.json 2> $O/pmc_mfma.err
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_gui -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-gemm-ceiling > $O/pmc_gui.json 2> $O/pmc_gui.err
# 4. other configurations, un-profiled lines
for c in "wan1.3b-81f bf16" "wan14b-81f bf16" "wan14b-81f fp8" "hunyuan-129f bf16" "hunyuan-129f fp8" "hunyuan-117f bf16"; do set -- $c; python3 $R/bench.py --config $1 --dtype $2 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_$1_$2.json 2>> $O/bench_cfg.err; done
python3 $R/bench.py --mix sparse-heavy --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_hunyuan129f_sparse_heavy.json 2>> $O/bench_cfg.err
python3 $R/bench.py --mix all-full --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-ceiling > $O/bench_hunyuan129f_allfull.json 2>> $O/bench_cfg.err
# 5. fp8 on its configuration under rocprofv3 (kernel stats + MFMA-busy / clock PMC)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fp8 -- python3 $R/bench.py --config wan14b-81f --dtype fp8 --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-ceiling > $O/bench_wan14b-81f_fp8_under_rocprof.json 2> $O/stats_fp8.err
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_mfma_fp8 -- python3 $R/bench.py --config wan14b-81f --dtype fp8 --steps 1 --warmup 0 --no-cpu-baseline --no-gemm-ceiling > /dev/null 2> $O/pmc_mfma_fp8.err
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_gui_fp8 -- python3 $R/bench.py --config wan14b-81f --dtype fp8 --steps 1 --warmup 0 --no-cpu-baseline --no-gemm-ceiling > /dev/null 2> $O/pmc_gui_fp8.err
# 6. producer pass / quantiser bandwidth, plain and under --stats
python3 $R/tools/bench_norm_rope.py > $O/norm_rope_bandwidth.txt 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_nr -- python3 $R/tools/bench_norm_rope.py > /dev/null 2> $O/stats_nr.err
python3 $R/tools/bench_gemm_ceiling.py > $O/gemm_ceiling.txt 2>&1
cd $R && python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/pmc_gui --match attn --json $O/pmc_hunyuan.json > /dev/null
python3 tools/pmc_summary.py $O/pmc_mfma_fp8 $O/pmc_gui_fp8 --match attn8 --json $O/pmc_fp8.json > /dev/null
ls $O | head -50
