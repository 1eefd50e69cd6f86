# GPU call 10: (a) parity of the re-swizzled two-wave K1 backward and K2 patch gradient; (b) SQ pass over the eager headline step
# (LDS conflict share of the K1 backward); (c) cycle stamps inside the fused patch-layer kernel (variant build)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_coattn_mcat.py tests/test_gpu_coattn_nacagat.py tests/test_gpu_patch_coattn.py tests/test_gpu_models.py -m gpu -q > gpurun_out/r03_t10.log 2>&1; rc=$?
echo "tests rc=$rc"; grep -E "passed|failed" gpurun_out/r03_t10.log | tail -2
if [ $rc -ne 0 ]; then grep -E "^FAILED|Error" gpurun_out/r03_t10.log | head; exit $rc; fi
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
rm -rf $R/gpurun_out/pmc_mcat_SQ
timeout -k 10 240 rocprofv3 --pmc $SQ -d $R/gpurun_out/pmc_mcat_SQ -o p --output-format csv -- python3 $R/bench.py --no-extras --no-cpu-baseline --no-graph --steps 6 --warmup 2 > $R/gpurun_out/pmc_mcat_SQ.log 2>&1; rc=$?
echo "pmc SQ rc=$rc"
cd $R
python tools/pmc_summarize.py gpurun_out/r03_pmc_mcat_sq2.json gpurun_out/pmc_mcat_SQ --match coattn_bwd8_kernel | cut -c1-700
timeout -k 10 120 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r03_b10.json 2>gpurun_out/r03_b10.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/r03_b10.json
cp multimodal_path_omic_amd/libmpo_hip.so /tmp/keep.so && cp multimodal_path_omic_amd/libmpo_hip_stamps.so multimodal_path_omic_amd/libmpo_hip.so
timeout -k 10 120 python tools/gpu_f1_stamps.py > gpurun_out/r03_f1_stamps.log 2>&1; echo "stamps rc=$?"
cp /tmp/keep.so multimodal_path_omic_amd/libmpo_hip.so
tail -60 gpurun_out/r03_f1_stamps.log
exit 0
