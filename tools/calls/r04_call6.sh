# r04 call 6: the skewed two-stream patch-layer kernel (patch_fc_fwd.hip) + K1 forward as its own launch: parity, timing
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_patch_coattn.py -m gpu -x -q > $O/c6_f1_tests.log 2>&1; rc=$?
echo "f1 tests rc=$rc"; tail -12 $O/c6_f1_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 200 python tools/gpu_time_f1.py > $O/c6_time.log 2>&1 || exit 1
K1=1 timeout -k 10 200 python tools/gpu_time_f1.py >> $O/c6_time.log 2>&1 || exit 1
DROP=0 timeout -k 10 200 python tools/gpu_time_f1.py >> $O/c6_time.log 2>&1 || exit 1
timeout -k 10 120 python tools/gpu_time_wgrad.py >> $O/c6_time.log 2>&1
grep -v amdgpu.ids $O/c6_time.log
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > $O/c6_bench.json 2> $O/c6_bench.err || { tail -5 $O/c6_bench.err; exit 1; }
python - <<PY
import json
d=json.load(open('$O/c6_bench.json'))
print('bench', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])
PY
