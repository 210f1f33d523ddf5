# Round 3, second batch: slot groups on alternating streams, coreset key order A/B, processor-level lines.
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun (it exports GRAFT_REPO_ROOT)}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3/b
rm -rf "$O" && mkdir -p "$O"
cd "$R"
B="python3 bench.py --no-cpu-baseline --no-gemm-ceiling"
for v in "g1 " "g2 --sp-groups 2" "g5 --sp-groups 5"; do
  set -- $v; n=$1; shift
  $B --config wan14b-81f --dtype fp8 --emulate-rank 8 --steps 4 --warmup 1 "$@" > $O/wan14b_fp8_rank8_$n.json 2>> $O/err.txt
done
VORTA_SP_GROUP_STREAMS=0 $B --config wan14b-81f --dtype fp8 --emulate-rank 8 --steps 4 --warmup 1 --sp-groups 2 > $O/wan14b_fp8_rank8_g2_onestream.json 2>> $O/err.txt
$B --config hunyuan-129f --emulate-rank 8 --steps 4 --warmup 1 --sp-groups 3 > $O/hunyuan_fp16_rank8_g3.json 2>> $O/err.txt
# coreset key order: serial launches so the coreset launch has its own line
$B --config hunyuan-129f --steps 2 --warmup 1 --experts serial > $O/hunyuan_fp16_serial_kv_group.json 2>> $O/err.txt
VORTA_CORESET_KV_ORDER=packed $B --config hunyuan-129f --steps 2 --warmup 1 --experts serial > $O/hunyuan_fp16_serial_kv_packed.json 2>> $O/err.txt
$B --config hunyuan-129f --steps 2 --warmup 1 > $O/hunyuan_fp16_fused_kv_group.json 2>> $O/err.txt
VORTA_CORESET_KV_ORDER=packed $B --config hunyuan-129f --steps 2 --warmup 1 > $O/hunyuan_fp16_fused_kv_packed.json 2>> $O/err.txt
# the production call path
$B --config hunyuan-129f --steps 2 --warmup 1 --level processor > $O/hunyuan_fp16_processor.json 2>> $O/err.txt
$B --config wan14b-81f --steps 2 --warmup 1 --level processor > $O/wan14b_bf16_processor.json 2>> $O/err.txt
$B --config wan14b-81f --dtype fp8 --steps 2 --warmup 1 --level processor > $O/wan14b_fp8_processor.json 2>> $O/err.txt
python3 - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r3/b/*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["dtype"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["share_of_step"], d["roofline"]["avg_launch_ms"])
        if "serial" in f:
            for k, v in d["per_launch"].items():
                print("    ", k, v)
    except Exception as e:
        print(os.path.basename(f), "-", str(e)[:60])
PY
tail -5 $O/err.txt
