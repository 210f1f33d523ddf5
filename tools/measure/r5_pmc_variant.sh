set -eux
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5/pmc_var; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
NB="--no-cpu-baseline --no-gemm-ceiling"
export VORTA_HIP_LIB=$R/vorta_amd/csrc/libvorta_hip_${VAR}.so
timeout -k 10 280 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $O/mfma -- python3 $R/bench.py --config wan14b-81f --dtype ${DT:-fp8pv} --steps 1 --warmup 0 $NB > /dev/null 2> $O/mfma.err
timeout -k 10 280 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/gui -- python3 $R/bench.py --config wan14b-81f --dtype ${DT:-fp8pv} --steps 1 --warmup 0 $NB > /dev/null 2> $O/gui.err
cd $R
python3 tools/pmc_summary.py $O/mfma $O/gui --match ${MATCH:-attn_mx} --json $O/pmc.json > $O/pmc.txt || true
find $O -name "*counter_collection.csv" -size +8M -delete || true
python3 - <<'PY'
import json,os
d=json.load(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r5/pmc_var/pmc.json'))
for k,v in d.items():
    c=v['counters']; wc=c['SQ_WAVE_CYCLES']['per_launch']
    print(k[:50], v['derived'], round(c['SQ_WAVE_CYCLES']['avg_duration_ms'],2),'ms')
    print('  valu active %.3f  wait_any %.3f  wait_inst %.3f'%(c['SQ_ACTIVE_INST_VALU']['per_launch']/wc, c['SQ_WAIT_ANY']['per_launch']/wc, c['SQ_WAIT_INST_ANY']['per_launch']/wc))
PY
