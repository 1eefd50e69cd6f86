# GPU call 4: fp16-split fp32 patch layer, fused K2 patch-side gradient, A/B bars: parity, then the bench line with extras
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_coattn_mcat.py tests/test_gpu_coattn_nacagat.py tests/test_gpu_patch_fc_f32.py tests/test_gpu_models.py tests/test_gpu_bag_selfattn.py tests/test_gpu_cohort.py tests/test_gpu_graph.py tests/test_gpu_dp.py -m gpu -q -rA > gpurun_out/r03_t4.log 2>&1; rc=$?
echo "tests rc=$rc"; grep -E "passed|failed" gpurun_out/r03_t4.log | tail -3
if [ $rc -gt 1 ]; then exit $rc; fi
grep -E "^FAILED" gpurun_out/r03_t4.log | head -30
grep -n "AssertionError: (" gpurun_out/r03_t4.log | head
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r03_b4.json 2> gpurun_out/r03_b4.err; rc=$?
echo "bench rc=$rc"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_b4.json'))
def show(n,r):
    rf=r.get('roofline') or {}
    print(n, r.get('value'), r.get('ms_per_step'), rf.get('avg_launch_us'), rf.get('frac'), r.get('error'))
show('headline',d)
for k,v in d.get('extra',{}).items(): show(k,v)
PY
if [ $rc -ne 0 ]; then exit $rc; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_b4_nacagat -o b4 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --model nacagat > $GRAFT_REPO_ROOT/gpurun_out/r03_b4_nacagat.json 2> $GRAFT_REPO_ROOT/gpurun_out/r03_b4_nacagat.err; rc=$?
echo "rocprof nacagat rc=$rc"
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/prof_b4_nacagat/*kernel_stats.csv 2>/dev/null | head -1); echo $f; head -16 $f | cut -c1-130
if [ $rc -ne 0 ]; then exit $rc; fi
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_b4_100k -o b4 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --patches 100000 --dtype f32 --window 8 --steps 8 > $GRAFT_REPO_ROOT/gpurun_out/r03_b4_100k.json 2> $GRAFT_REPO_ROOT/gpurun_out/r03_b4_100k.err; rc=$?
echo "rocprof 100k rc=$rc"
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/prof_b4_100k/*kernel_stats.csv 2>/dev/null | head -1); echo $f; head -8 $f | cut -c1-130
exit $rc
