# Round 4: is the K/V stream's latency exposed in the product loop now?  Same-box alternating A/B of the fused fp16 layer
# kernel: product (ring 2, K prefetch 2) vs ring 3 + K prefetch 1 (two steps of stream in flight) vs K prefetch 1 alone.
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4/ab_ring
rm -rf $O && mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-gemm-ceiling --steps 2 --warmup 1"
for rep in 1 2 3; do
  for v in base _r3k1 _k1; do
    s=$v; [ "$v" = base ] && s=""
    VORTA_HIP_LIB=$PWD/vorta_amd/csrc/libvorta_hip$s.so $B > $O/${v}_fp16_$rep.json 2>> $O/err.txt
  done
done
for f in $O/*.json; do python3 -c "
import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['switches']['library'][-40:])"; done | tee $O/summary.txt
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1 || (tail -60 $O/pytest_gpu.txt; exit 1)
tail -3 $O/pytest_gpu.txt
