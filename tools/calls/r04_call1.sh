# r04 call 1: the rebuilt f1 kernel (output-stationary 256 x 256 blocks): its parity tests, then its stand-alone timing.
mkdir -p gpurun_out/r04
O=gpurun_out/r04
timeout -k 10 500 python -m pytest tests/test_gpu_patch_coattn.py -m gpu -x -q > $O/c1_f1_tests.log 2>&1; rc=$?
echo "f1 tests rc=$rc"; tail -15 $O/c1_f1_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 200 python tools/gpu_time_f1.py > $O/c1_f1_time.log 2>&1 && cat $O/c1_f1_time.log
DROP=0 timeout -k 10 200 python tools/gpu_time_f1.py >> $O/c1_f1_time.log 2>&1 && tail -1 $O/c1_f1_time.log
