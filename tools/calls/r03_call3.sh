# GPU call 3: two-wave K1 backward + fp32 patch-layer kernels: parity, then bench (headline + cfg 5) and rocprofv3 kernel stats
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_coattn_mcat.py tests/test_gpu_patch_fc_f32.py tests/test_gpu_patch_coattn.py tests/test_gpu_models.py tests/test_gpu_graph.py tests/test_gpu_bag_selfattn.py tests/test_gpu_cohort.py -m gpu -q -rA > gpurun_out/r03_t3.log 2>&1; rc=$?
echo "tests rc=$rc"; grep -E "passed|failed" gpurun_out/r03_t3.log | tail -3
if [ $rc -gt 1 ]; then exit $rc; fi
grep -E "^(FAILED|E  )" gpurun_out/r03_t3.log | head -30
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03_b3.json 2> gpurun_out/r03_b3.err; rc=$?
echo "bench rc=$rc"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_b3.json'))
def show(n,r):
    rf=r.get('roofline') or {}
    print(n, r.get('value'), r.get('ms_per_step'), rf.get('avg_launch_us'), rf.get('frac'), r.get('error'))
show('headline',d)
for k,v in d.get('extra',{}).items(): show(k,v)
PY
if [ $rc -ne 0 ]; then exit $rc; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_b3 -o b3 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r03_b3_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r03_b3_prof.err; rc=$?
echo "rocprof rc=$rc"
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/prof_b3/*kernel_stats.csv 2>/dev/null | head -1); echo $f; head -8 $f | cut -c1-150
if [ $rc -ne 0 ]; then exit $rc; fi
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_b3_100k -o b3 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --patches 100000 --dtype f32 --window 8 --steps 8 > $GRAFT_REPO_ROOT/gpurun_out/r03_b3_100k.json 2> $GRAFT_REPO_ROOT/gpurun_out/r03_b3_100k.err; rc=$?
echo "rocprof 100k rc=$rc"
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/prof_b3_100k/*kernel_stats.csv 2>/dev/null | head -1); echo $f; head -10 $f | cut -c1-150
exit $rc
