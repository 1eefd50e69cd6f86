O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_patch_fc_f32.py tests/test_gpu_models.py -m gpu -x -q > $O/c27_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -2 $O/c27_tests.log
if [ $rc -ne 0 ]; then grep -E "^E |FAILED" $O/c27_tests.log | head -20; exit $rc; fi
bash tools/calls/r04_prof.sh f32_swz --patches 100000 --dtype f32 --window 8 --steps 8 | head -8
