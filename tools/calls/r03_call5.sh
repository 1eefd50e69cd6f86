# GPU call 5: re-run the failing groups (fp16 RNE split, A/B colsum through a bucket, multi-tile K2 patch gradient)
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_coattn_mcat.py tests/test_gpu_coattn_nacagat.py tests/test_gpu_patch_fc_f32.py tests/test_gpu_models.py tests/test_gpu_bag_selfattn.py -m gpu -q -rA > gpurun_out/r03_t5.log 2>&1; rc=$?
echo "tests rc=$rc"; grep -E "passed|failed" gpurun_out/r03_t5.log | tail -3
grep -E "^FAILED" gpurun_out/r03_t5.log | head -30
grep -n "AssertionError" gpurun_out/r03_t5.log | head
exit 0
