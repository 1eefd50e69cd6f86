# Round 4 final, part 1: the whole GPU suite on one box (one process).  Output: gpurun_out/final/gpu_tests.log
mkdir -p gpurun_out/final
O=gpurun_out/final
timeout -k 10 1100 python -m pytest tests -m gpu -q -rA > $O/gpu_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; grep -E "passed|failed" $O/gpu_tests.log | tail -2
grep -E "^FAILED|^ERROR" $O/gpu_tests.log | head
exit $rc
