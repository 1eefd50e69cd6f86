# r04 call 16: the larger cohort fixture (80 slides, 3 epochs) in fp32 and bf16 storage; whole-model parity at 100 000 fp32 patches
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest "tests/test_gpu_cohort.py" -k bf16 -m gpu -q -s > $O/c16_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; grep -E "^\[cohort|^\[cfg5|passed|failed|^E  " $O/c16_tests.log | grep -v f32 | head -60
exit $rc
