set -eux
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3/mx
mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-gemm-ceiling --steps 2 --warmup 1"
$B --config wan14b-81f --dtype bf16 > $O/wan_bf16.json 2>> $O/err.txt
$B --config wan14b-81f --dtype fp8pv > $O/wan_fp8pv.json 2>> $O/err.txt
$B --config wan14b-81f --dtype fp8 > $O/wan_fp8.json 2>> $O/err.txt
$B --config hunyuan-129f --dtype bf16 > $O/hy_bf16.json 2>> $O/err.txt
$B --config hunyuan-129f --dtype fp8pv > $O/hy_fp8pv.json 2>> $O/err.txt
$B --config wan14b-81f --dtype fp8pv --experts serial > $O/wan_fp8pv_serial.json 2>> $O/err.txt
for n in wan_bf16 wan_fp8pv wan_fp8 hy_bf16 hy_fp8pv wan_fp8pv_serial; do python3 -c "
import json; d=json.loads(open('$O/$n.json').read().strip().splitlines()[-1]); print('$n', d['ms_per_step'], d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'], d['roofline']['share_of_step'])"; done
tail -3 $O/err.txt
