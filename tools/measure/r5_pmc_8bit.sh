set -eux
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5/pmc_fp8pv; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
NB="--no-cpu-baseline --no-gemm-ceiling"
for dt in fp8pv fp8; do
timeout -k 10 280 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $O/mfma_$dt -- python3 $R/bench.py --config wan14b-81f --dtype $dt --steps 1 --warmup 0 $NB > /dev/null 2> $O/mfma_$dt.err
timeout -k 10 280 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/gui_$dt -- python3 $R/bench.py --config wan14b-81f --dtype $dt --steps 1 --warmup 0 $NB > /dev/null 2> $O/gui_$dt.err
done
cd $R
python3 tools/pmc_summary.py $O/mfma_fp8pv $O/gui_fp8pv --match attn_mx --json $O/pmc_fp8pv.json > $O/pmc_fp8pv.txt || true
python3 tools/pmc_summary.py $O/mfma_fp8 $O/gui_fp8 --match attn8 --json $O/pmc_fp8.json > $O/pmc_fp8.txt || true
find $O -name "*counter_collection.csv" -size +8M -delete || true
cat $O/pmc_fp8pv.txt $O/pmc_fp8.txt | tail -40
