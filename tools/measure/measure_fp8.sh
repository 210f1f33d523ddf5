# The fp8 part of tools/measure_round.sh + the emulated rank of 8, for re-taking those artefacts after an fp8-only change.
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun (it exports GRAFT_REPO_ROOT)}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2/f8
rm -rf "$O" && mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for c in "wan14b-81f bf16" "wan14b-81f fp8" "hunyuan-129f bf16" "hunyuan-129f fp8"; do set -- $c; python3 $R/bench.py --config $1 --dtype $2 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_$1_$2.json 2>> $O/err.txt; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fp8 -- python3 $R/bench.py --config wan14b-81f --dtype fp8 --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-ceiling > $O/bench_wan14b-81f_fp8_under_rocprof.json 2> $O/stats_fp8.err
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_mfma_fp8 -- python3 $R/bench.py --config wan14b-81f --dtype fp8 --steps 1 --warmup 0 --no-cpu-baseline --no-gemm-ceiling > /dev/null 2> $O/pmc_mfma_fp8.err
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_gui_fp8 -- python3 $R/bench.py --config wan14b-81f --dtype fp8 --steps 1 --warmup 0 --no-cpu-baseline --no-gemm-ceiling > /dev/null 2> $O/pmc_gui_fp8.err
cd $R
python3 bench.py --config wan14b-81f --dtype fp8 --emulate-rank 8 --steps 3 --warmup 1 --no-cpu-baseline --no-gemm-ceiling > $O/emulated_rank_of_8_wan14b_fp8.json 2>> $O/err.txt
python3 bench.py --config wan14b-81f --dtype bf16 --emulate-rank 8 --steps 3 --warmup 1 --no-cpu-baseline --no-gemm-ceiling > $O/emulated_rank_of_8_wan14b_bf16.json 2>> $O/err.txt
python3 tools/pmc_summary.py $O/pmc_mfma_fp8 $O/pmc_gui_fp8 --match attn8 --json $O/pmc_fp8.json > /dev/null
python3 - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r2/f8/*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["dtype"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["share_of_step"], d["roofline"]["avg_launch_ms"])
    except Exception as e:
        print(os.path.basename(f), "-", str(e)[:60])
PY
