# One rank's share of a P-GPU Ulysses step on one GPU (bench.py --emulate-rank), same box as the 1-GPU line.
# Run on the GPU box:  bash tools/measure_emulate.sh   (writes gpurun_out/r2/e/*.json; copy what is kept into profiles/)
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun (it exports GRAFT_REPO_ROOT)}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2/e
mkdir -p $O
cd $R
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gemm-ceiling > $O/emulated_rank_same_box_1gpu.json 2> $O/err.txt
for P in 2 4 8; do python3 bench.py --emulate-rank $P --steps 3 --warmup 1 --no-cpu-baseline --no-gemm-ceiling > $O/emulated_rank_of_$P.json 2>> $O/err.txt; done
for d in bf16 fp8; do python3 bench.py --config wan14b-81f --dtype $d --emulate-rank 8 --steps 3 --warmup 1 --no-cpu-baseline --no-gemm-ceiling > $O/emulated_rank_of_8_wan14b_$d.json 2>> $O/err.txt; done
VORTA_SP_STAGING=torch python3 bench.py --emulate-rank 8 --steps 3 --warmup 1 --no-cpu-baseline --no-gemm-ceiling > $O/emulated_rank_of_8_torch_staging.json 2>> $O/err.txt
python3 - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r2/e/*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d["dtype"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["share_of_step"])
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
