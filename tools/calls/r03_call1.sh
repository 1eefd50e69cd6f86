# GPU call 1 of round 3: full GPU test suite with the tightened bars, baseline bench on this box, overlap probe.
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -q -rA > gpurun_out/r03_t1.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -3 gpurun_out/r03_t1.log
if [ $rc -gt 1 ]; then exit $rc; fi          # killed / timed out: no further GPU step in this call
timeout -k 10 240 python bench.py > gpurun_out/r03_base_bench.json 2> gpurun_out/r03_base_bench.err; rc=$?
echo "bench rc=$rc"; if [ $rc -ne 0 ]; then tail -5 gpurun_out/r03_base_bench.err; exit $rc; fi
timeout -k 10 300 python tools/gpu_probe_overlap.py > gpurun_out/r03_overlap.log 2>&1; rc=$?
echo "probe rc=$rc"; tail -8 gpurun_out/r03_overlap.log
exit $rc
