O=gpurun_out/r04; mkdir -p $O
P=multimodal_path_omic_amd
cp $P/libmpo_hip.so /tmp/keep.so
cp $P/libmpo_hip_w4.so $P/libmpo_hip.so
timeout -k 10 300 python -m pytest tests/test_gpu_patch_coattn.py -m gpu -x -q > $O/c33_tests.log 2>&1; rc=$?
echo "w4 tests rc=$rc"; tail -2 $O/c33_tests.log
if [ $rc -ne 0 ]; then grep -E "^E |FAILED" $O/c33_tests.log | head; cp /tmp/keep.so $P/libmpo_hip.so; exit $rc; fi
for v in base w4 base w4 base w4; do
  cp $P/libmpo_hip_$v.so $P/libmpo_hip.so
  timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline > $O/c33_$v.json 2> $O/c33_$v.err || { echo "$v failed"; tail -5 $O/c33_$v.err; cp /tmp/keep.so $P/libmpo_hip.so; exit 1; }
  python - <<PY
import json
d=json.load(open('$O/c33_$v.json')); r=d['roofline']; print('$v', d['value'], d['ms_per_step'], r.get('avg_launch_us'))
PY
done
cp $P/libmpo_hip_w4stamps.so $P/libmpo_hip.so
timeout -k 10 200 python tools/gpu_f1_stamps.py 2>&1 | tail -16
cp /tmp/keep.so $P/libmpo_hip.so
