# Round 3, third batch: processor-level lines after the slot spread; structured-input accuracy of the e4m3 path.
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun (it exports GRAFT_REPO_ROOT)}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3/c
rm -rf "$O" && mkdir -p "$O"
cd "$R"
B="python3 bench.py --no-cpu-baseline --no-gemm-ceiling"
$B --config hunyuan-129f --steps 2 --warmup 1 > $O/hunyuan_fp16_attention.json 2>> $O/err.txt
$B --config hunyuan-129f --steps 2 --warmup 1 --level processor > $O/hunyuan_fp16_processor.json 2>> $O/err.txt
$B --config wan14b-81f --steps 2 --warmup 1 --level processor > $O/wan14b_bf16_processor.json 2>> $O/err.txt
$B --config wan14b-81f --dtype fp8 --steps 2 --warmup 1 --level processor > $O/wan14b_fp8_processor.json 2>> $O/err.txt


python3 - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r3/c/*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["dtype"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["share_of_step"], d["roofline"]["avg_launch_ms"], d["config"].get("ms_per_layer"))
    except Exception as e:
        print(os.path.basename(f), "-", str(e)[:60])
PY

