# Round 4: part D of tools/probe_mfma_shape.hip (the K/V tile stream beside the step), then the GPU suite of the working tree.
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun (it exports GRAFT_REPO_ROOT)}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4/probe_dma
rm -rf "$O" && mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -mllvm -enable-post-misched=0 -w $R/tools/probe_mfma_shape.hip -o /tmp/probe_shape
( while true; do echo "t $(date +%s.%N)"; rocm-smi --showpower --showclocks --json 2>/dev/null | head -c 2000; echo; sleep 0.25; done ) > $O/smi_samples.txt 2>&1 &
SAMPLER=$!
timeout -k 10 300 /tmp/probe_shape 8 2 > $O/probe.txt 2>&1 || true
kill $SAMPLER || true
cd $R
python3 tools/smi_phases.py $O/probe.txt $O/smi_samples.txt > $O/probe_with_power.txt || true
cat $O/probe_with_power.txt
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1 || (tail -60 $O/pytest_gpu.txt; exit 1)
tail -3 $O/pytest_gpu.txt
