# Round 3: one rank's share of the 8-GPU fp8 step (BASELINE configs[4]) with the exchange variants, next to the 1-GPU
# lines of the same box.  Run on the GPU box:  bash tools/measure_r3_sp.sh   (writes gpurun_out/r3/sp/*.json)
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun (it exports GRAFT_REPO_ROOT)}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3/sp
rm -rf "$O" && mkdir -p "$O"
cd "$R"
B="python3 bench.py --no-cpu-baseline --no-gemm-ceiling"
$B --config wan14b-81f --dtype fp8 --steps 3 --warmup 1 > $O/wan14b_fp8_p1.json 2>> $O/err.txt
$B --config wan14b-81f --dtype bf16 --steps 2 --warmup 1 > $O/wan14b_bf16_p1.json 2>> $O/err.txt
for v in "g1_v16 --no-v-wire" "g1_v8 " "g2_v8 --sp-groups 2" "g2_v16 --sp-groups 2 --no-v-wire" "g5_v8 --sp-groups 5"; do
  set -- $v; n=$1; shift
  $B --config wan14b-81f --dtype fp8 --emulate-rank 8 --steps 4 --warmup 1 "$@" > $O/wan14b_fp8_rank8_$n.json 2>> $O/err.txt
done
$B --config wan14b-81f --dtype bf16 --emulate-rank 8 --steps 4 --warmup 1 > $O/wan14b_bf16_rank8_g1.json 2>> $O/err.txt
$B --config wan14b-81f --dtype bf16 --emulate-rank 8 --steps 4 --warmup 1 --sp-groups 2 > $O/wan14b_bf16_rank8_g2.json 2>> $O/err.txt
$B --config hunyuan-129f --steps 2 --warmup 1 > $O/hunyuan_fp16_p1.json 2>> $O/err.txt
$B --config hunyuan-129f --emulate-rank 8 --steps 4 --warmup 1 > $O/hunyuan_fp16_rank8_g1.json 2>> $O/err.txt
$B --config hunyuan-129f --dtype fp8 --emulate-rank 8 --steps 4 --warmup 1 > $O/hunyuan_fp8_rank8_g1_v8.json 2>> $O/err.txt
python3 tools/dbg/fp8_quant_mall.py > $O/quant_mall.txt 2>> $O/err.txt
python3 - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r3/sp/*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["dtype"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["share_of_step"], d["roofline"]["avg_launch_ms"])
    except Exception as e:
        print(os.path.basename(f), "-", str(e)[:60])
PY
cat $O/quant_mall.txt
