mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -rA > gpurun_out/r03_t8.log 2>&1; rc=$?
echo "tests rc=$rc"; grep -E "passed|failed" gpurun_out/r03_t8.log | tail -3
grep -E "^FAILED" gpurun_out/r03_t8.log | head -30
grep -n "AssertionError\|Error:" gpurun_out/r03_t8.log | head -20
grep -E "big nacagat|K2 grads|K2 map" gpurun_out/r03_t8.log | head -40
exit 0
