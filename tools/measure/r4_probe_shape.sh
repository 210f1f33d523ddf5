# Round 4, step 0 of VERDICT r03 item 1: MFMA shape / LDS-read energy probe (tools/probe_mfma_shape.hip) with the driver's
# clock and socket power sampled beside it; then the GPU test suite and the default bench line of the unchanged tree.
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun (it exports GRAFT_REPO_ROOT)}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4/probe
rm -rf "$O" && mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -mllvm -enable-post-misched=0 -w $R/tools/probe_mfma_shape.hip -o /tmp/probe_shape
( while true; do echo "t $(date +%s.%N)"; rocm-smi --showpower --showclocks --json 2>/dev/null | head -c 2000; echo; sleep 0.25; done ) > $O/smi_samples.txt 2>&1 &
SAMPLER=$!
timeout -k 10 600 /tmp/probe_shape 7 2 > $O/probe.txt 2>&1 || true
kill $SAMPLER || true
tail -80 $O/probe.txt
cd $R
python3 tools/smi_phases.py $O/probe.txt $O/smi_samples.txt > $O/probe_with_power.txt || true
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1 || (tail -30 $O/pytest_gpu.txt; exit 1)
tail -3 $O/pytest_gpu.txt
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || tail -5 $O/bench_default.err
cat $O/bench_default.json
