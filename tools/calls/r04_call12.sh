O=gpurun_out/r04; mkdir -p $O
P=multimodal_path_omic_amd
timeout -k 10 500 python -m pytest tests/test_gpu_patch_coattn.py -m gpu -x -q > $O/c12_f1_tests.log 2>&1; rc=$?
echo "f1 tests rc=$rc"; tail -3 $O/c12_f1_tests.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $O/c12_f1_tests.log | head -20; exit $rc; fi
cp $P/libmpo_hip.so /tmp/keep.so
rm -f $O/c12.log
for v in keep dma; do
  if [ $v = keep ]; then cp /tmp/keep.so $P/libmpo_hip.so; else cp $P/libmpo_hip_$v.so $P/libmpo_hip.so; fi
  echo "== $v" >> $O/c12.log
  timeout -k 10 120 python tools/gpu_time_f1.py >> $O/c12.log 2>&1 || exit 1
done
cp $P/libmpo_hip_stamps.so $P/libmpo_hip.so
timeout -k 10 120 python tools/gpu_f1_stamps.py >> $O/c12.log 2>&1
cp /tmp/keep.so $P/libmpo_hip.so
grep -v amdgpu.ids $O/c12.log
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > $O/c12_bench.json 2> $O/c12_bench.err || { tail -5 $O/c12_bench.err; exit 1; }
python - <<PY
import json
d=json.load(open('$O/c12_bench.json'))
print('bench', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])
PY
