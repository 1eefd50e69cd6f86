O=gpurun_out/r04; mkdir -p $O
P=multimodal_path_omic_amd
timeout -k 10 500 python -m pytest tests/test_gpu_patch_coattn.py -m gpu -x -q > $O/c9_f1_tests.log 2>&1; rc=$?
echo "f1 tests rc=$rc"; tail -3 $O/c9_f1_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
cp $P/libmpo_hip.so /tmp/keep.so
rm -f $O/c9.log
for v in keep dma xonly mmaonly; do
  if [ $v = keep ]; then cp /tmp/keep.so $P/libmpo_hip.so; else cp $P/libmpo_hip_$v.so $P/libmpo_hip.so; fi
  echo "== $v" >> $O/c9.log
  timeout -k 10 120 python tools/gpu_time_f1.py >> $O/c9.log 2>&1 || exit 1
done
cp /tmp/keep.so $P/libmpo_hip.so
K1=1 timeout -k 10 200 python tools/gpu_time_f1.py >> $O/c9.log 2>&1 || exit 1
grep -v amdgpu.ids $O/c9.log
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > $O/c9_bench.json 2> $O/c9_bench.err || { tail -5 $O/c9_bench.err; exit 1; }
python - <<PY
import json
d=json.load(open('$O/c9_bench.json'))
print('bench', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])
PY
