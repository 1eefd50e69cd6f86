# GPU call 2: the two-waves-per-SIMD K1 backward: parity, then its time inside the bench step (rocprofv3 kernel stats)
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_coattn_mcat.py tests/test_gpu_patch_coattn.py tests/test_gpu_models.py tests/test_gpu_graph.py -m gpu -q -x -rA > gpurun_out/r03_t2.log 2>&1; rc=$?
echo "tests rc=$rc"; grep -E "passed|failed|Error" gpurun_out/r03_t2.log | tail -5
if [ $rc -ne 0 ]; then grep -E "^E " gpurun_out/r03_t2.log | head -20; exit $rc; fi
timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r03_b2.json 2> gpurun_out/r03_b2.err; rc=$?
echo "bench rc=$rc"; cat gpurun_out/r03_b2.json | cut -c1-400
if [ $rc -ne 0 ]; then exit $rc; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_b2 -o b2 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r03_b2_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r03_b2_prof.err; rc=$?
echo "rocprof rc=$rc"
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/prof_b2/*kernel_stats.csv 2>/dev/null | head -1); echo $f; head -12 $f | cut -c1-160
exit $rc
