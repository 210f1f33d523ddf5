# Round 3, late: refresh of the lines the last kernel changes touch (e4m3 recipe, quantiser, producer pass), final binary, one box
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun (it exports GRAFT_REPO_ROOT)}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3/late
rm -rf "$O" && mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_hunyuan129f_uniform.json 2> $O/bench_default.err || tail -5 $O/bench_default.err
for c in "wan14b-81f fp8" "hunyuan-129f fp8" "wan14b-81f fp8pv" "wan14b-81f bf16"; do set -- $c; python3 $R/bench.py --config $1 --dtype $2 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_$1_$2.json 2>> $O/bench_cfg.err; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fp8 -- python3 $R/bench.py --config wan14b-81f --dtype fp8 --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-ceiling > $O/bench_wan14b-81f_fp8_under_rocprof.json 2> $O/stats_fp8.err
python3 $R/bench.py --config wan14b-81f --dtype fp8 --emulate-rank 8 --steps 4 --warmup 1 --no-gemm-ceiling > $O/rank_of_8_wan14b_fp8.json 2>> $O/bench_cfg.err
python3 $R/bench.py --level processor --steps 2 --warmup 1 --no-gemm-ceiling > $O/bench_hunyuan129f_processor_level.json 2>> $O/bench_cfg.err
python3 $R/bench.py --config wan14b-81f --dtype fp8 --level processor --steps 2 --warmup 1 --no-gemm-ceiling > $O/bench_wan14b_fp8_processor_level.json 2>> $O/bench_cfg.err
python3 - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r3/late/*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(os.path.basename(f), d["dtype"], d["ms_per_step"], r["frac"], r["share_of_step"], r["avg_launch_ms"], r.get("library_gemm_tflops"), r.get("frac_of_library_gemm"))
    except Exception as e:
        print(os.path.basename(f), "-", str(e)[:60])
PY
find $O/stats_fp8 -name "*kernel_stats.csv" | head -2
