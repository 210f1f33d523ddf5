# Round 3: L2-side traffic (FETCH_SIZE / WRITE_SIZE, separate passes, no trace domains) of the dominant kernel for every
# workload bench.py is run on; tools/pmc_traffic_table.py folds them into profiles/r03_pmc_traffic.json (read by bench.py).
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun (it exports GRAFT_REPO_ROOT)}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3/pmc
rm -rf "$O" && mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
i=0
while read -r cfg mix dt; do
  i=$((i+1))
  tag="${cfg}_${mix}_${dt}"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 280 rocprofv3 --pmc $c --output-format csv -d $O/$tag/$c -- python3 $R/bench.py --config $cfg --mix $mix --dtype $dt --steps 1 --warmup 0 --no-cpu-baseline --no-gemm-ceiling > $O/$tag.$c.json 2> $O/$tag.$c.err
    echo "done $tag $c"
  done
done <<'LIST'
hunyuan-129f uniform fp16
hunyuan-129f uniform bf16
hunyuan-129f uniform fp8
hunyuan-129f sparse-heavy fp16
hunyuan-129f all-full fp16
wan14b-81f uniform bf16
wan14b-81f uniform fp8
wan1.3b-81f uniform bf16
LIST
cd $R && python3 tools/pmc_traffic_table.py $O --json $O/r03_pmc_traffic.json
