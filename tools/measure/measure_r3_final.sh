# Round 3: the committed measurement set of the final binary (one box): the default bench line, the same command under
# rocprofv3 --kernel-trace --stats, the other configurations, the e4m3 line under --stats, MFMA-busy / clock PMC passes.
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun (it exports GRAFT_REPO_ROOT)}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3/m
rm -rf "$O" && mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_hunyuan129f_uniform.json 2> $O/bench_default.err || tail -5 $O/bench_default.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_hy -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-ceiling > $O/bench_hunyuan129f_uniform_under_rocprof.json 2> $O/stats_hy.err
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_mfma -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-gemm-ceiling > $O/pmc_mfma.json 2> $O/pmc_mfma.err
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_gui -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-gemm-ceiling > $O/pmc_gui.json 2> $O/pmc_gui.err
for c in "wan1.3b-81f bf16" "wan14b-81f bf16" "wan14b-81f fp8" "hunyuan-129f bf16" "hunyuan-129f fp8" "hunyuan-117f bf16"; do set -- $c; python3 $R/bench.py --config $1 --dtype $2 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_$1_$2.json 2>> $O/bench_cfg.err; done
python3 $R/bench.py --mix sparse-heavy --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_hunyuan129f_sparse_heavy.json 2>> $O/bench_cfg.err
python3 $R/bench.py --mix all-full --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-ceiling > $O/bench_hunyuan129f_allfull.json 2>> $O/bench_cfg.err
python3 $R/bench.py --steps 10 --warmup 1 --no-cpu-baseline > $O/bench_hunyuan129f_uniform_10steps.json 2>> $O/bench_cfg.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fp8 -- python3 $R/bench.py --config wan14b-81f --dtype fp8 --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-ceiling > $O/bench_wan14b-81f_fp8_under_rocprof.json 2> $O/stats_fp8.err
python3 $R/bench.py --level processor --steps 2 --warmup 1 --no-gemm-ceiling > $O/bench_hunyuan129f_processor_level.json 2>> $O/bench_cfg.err
python3 $R/bench.py --config wan14b-81f --dtype fp8 --emulate-rank 8 --steps 4 --warmup 1 --no-gemm-ceiling > $O/rank_of_8_wan14b_fp8.json 2>> $O/bench_cfg.err
python3 $R/bench.py --emulate-rank 8 --steps 4 --warmup 1 --no-gemm-ceiling > $O/rank_of_8_hunyuan_fp16.json 2>> $O/bench_cfg.err
cd $R && python3 tools/pmc_summary.py $O/pmc_mfma $O/pmc_gui --match attn --json $O/pmc_hunyuan.json > /dev/null
python3 - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r3/m/*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["dtype"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["share_of_step"], d["roofline"]["avg_launch_ms"], d["roofline"].get("traffic"))
    except Exception as e:
        print(os.path.basename(f), "-", str(e)[:60])
PY
find $O/stats_hy -name "*kernel_stats.csv" | head -2
