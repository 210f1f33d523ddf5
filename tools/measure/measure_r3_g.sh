# Round 3: head placement -- equal head counts vs counts that follow the routes -- on the heaviest rank of 8 (no transfers).
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun (it exports GRAFT_REPO_ROOT)}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3/g
rm -rf "$O" && mkdir -p "$O"
cd "$R"
B="python3 bench.py --no-cpu-baseline --no-gemm-ceiling"
for pl in even uneven; do
  $B --config wan14b-81f --dtype fp8 --emulate-rank 8 --steps 4 --warmup 1 --placement $pl > $O/wan14b_fp8_rank8_$pl.json 2>> $O/err.txt
  $B --config wan14b-81f --dtype bf16 --emulate-rank 8 --steps 3 --warmup 1 --placement $pl > $O/wan14b_bf16_rank8_$pl.json 2>> $O/err.txt
  $B --config hunyuan-129f --mix sparse-heavy --emulate-rank 8 --steps 3 --warmup 1 --placement $pl > $O/hunyuan_fp16_sparseheavy_rank8_$pl.json 2>> $O/err.txt
  $B --config hunyuan-129f --emulate-rank 8 --steps 3 --warmup 1 --placement $pl > $O/hunyuan_fp16_uniform_rank8_$pl.json 2>> $O/err.txt
done
$B --config hunyuan-129f --mix sparse-heavy --steps 2 --warmup 1 > $O/hunyuan_fp16_sparseheavy_p1.json 2>> $O/err.txt
$B --config wan14b-81f --dtype fp8 --steps 2 --warmup 1 > $O/wan14b_fp8_p1.json 2>> $O/err.txt
$B --config wan14b-81f --dtype bf16 --steps 2 --warmup 1 > $O/wan14b_bf16_p1.json 2>> $O/err.txt
python3 - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r3/g/*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["dtype"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["share_of_step"], d["config"]["parallelism"][-60:])
    except Exception as e:
        print(os.path.basename(f), "-", str(e)[:60])
PY
tail -3 $O/err.txt
