O=gpurun_out/r04; mkdir -p $O
P=multimodal_path_omic_amd
cp $P/libmpo_hip.so /tmp/keep.so
for v in old new old new old new; do
  cp $P/libmpo_hip_$v.so $P/libmpo_hip.so
  timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline > $O/c24_$v.json 2> $O/c24_$v.err || { echo "$v failed"; tail -5 $O/c24_$v.err; cp /tmp/keep.so $P/libmpo_hip.so; exit 1; }
  python - <<PY
import json
d=json.load(open('$O/c24_$v.json')); r=d['roofline']; print('$v', d['value'], d['ms_per_step'], r.get('avg_launch_us'))
PY
done
cp /tmp/keep.so $P/libmpo_hip.so
python tools/gpu_time_wgrad.py
cp $P/libmpo_hip_old.so $P/libmpo_hip.so
python tools/gpu_time_wgrad.py
cp /tmp/keep.so $P/libmpo_hip.so
