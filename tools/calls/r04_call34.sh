O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_coattn_nacagat.py tests/test_gpu_models.py tests/test_gpu_cohort.py -m gpu -x -q > $O/c34_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -2 $O/c34_tests.log
if [ $rc -ne 0 ]; then grep -E "^E |FAILED" $O/c34_tests.log | head -20; exit $rc; fi
bash tools/calls/r04_prof.sh nacagat_sm --model nacagat | head -3
grep "gated_softmax" $O/nacagat_sm_step_timeline.txt | cut -c1-110
