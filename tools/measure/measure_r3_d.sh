# Round 3, fourth batch: DRAM-side activity of the fused launch; fp8 structured inputs with the smoothing experiment.
set -eux
: "${GRAFT_REPO_ROOT:?run through gpurun (it exports GRAFT_REPO_ROOT)}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3/d
rm -rf "$O" && mkdir -p "$O"
cd "$R"
rocm-smi --showmemuse --showuse --json > $O/smi_probe.json 2>&1 || true
python3 tools/umc_activity.py --json $O/umc_hunyuan_fp16.json > $O/umc_hunyuan_fp16.txt 2>> $O/err.txt || true
python3 tools/umc_activity.py --config wan14b-81f --dtype fp8 --json $O/umc_wan14b_fp8.json > $O/umc_wan14b_fp8.txt 2>> $O/err.txt || true
python3 tools/dbg/fp8_structured.py --geometry wan14b-81f > $O/fp8_structured_wan14b.txt 2>> $O/err.txt
python3 tools/dbg/fp8_structured.py --geometry hunyuan-129f > $O/fp8_structured_hunyuan.txt 2>> $O/err.txt
python3 -m pytest tests/test_hip_fp8.py -q -m gpu -s -k "routed_vs or operator_psnr" > $O/fp8_tests.txt 2>&1 || true
cat $O/smi_probe.json $O/umc_*.txt $O/fp8_structured_*.txt; grep "rel. Frobenius\|fp8 vs bf16" $O/fp8_tests.txt; tail -3 $O/err.txt
