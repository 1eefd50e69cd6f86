mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_patch_fc_f32.py tests/test_gpu_models.py -m gpu -q -k "fc_f32 or golden or big or small" > gpurun_out/r03_t15.log 2>&1; rc=$?
echo "tests rc=$rc"; grep -E "passed|failed" gpurun_out/r03_t15.log | tail -2
if [ $rc -ne 0 ]; then grep -E "^FAILED|Error" gpurun_out/r03_t15.log | head; exit $rc; fi
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_100k_b -o p --output-format csv -- python3 $R/bench.py --no-extras --no-cpu-baseline --patches 100000 --dtype f32 --window 8 --steps 8 > $R/gpurun_out/r03_b15_100k.json 2> $R/gpurun_out/r03_b15_100k.err; echo "rocprof rc=$?"
cd $R
cut -c1-200 gpurun_out/r03_b15_100k.json
head -6 gpurun_out/prof_100k_b/p_kernel_stats.csv | cut -c1-140
rm -f gpurun_out/prof_100k_b/p_kernel_trace.csv
