O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_coattn_mcat.py tests/test_gpu_models.py tests/test_gpu_patch_coattn.py -m gpu -x -q -s > $O/c29_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -2 $O/c29_tests.log; grep "K1 fp32 backward" $O/c29_tests.log
if [ $rc -ne 0 ]; then grep -E "^E |FAILED" $O/c29_tests.log | head -20; exit $rc; fi
