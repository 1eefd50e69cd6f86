O=gpurun_out/r04; mkdir -p $O
P=multimodal_path_omic_amd
cp $P/libmpo_hip.so /tmp/keep.so
for v in base wgissue bwd8rev epihi mainhi base wgissue bwd8rev epihi mainhi; do
  cp $P/libmpo_hip_$v.so $P/libmpo_hip.so
  timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline > $O/c21_$v.json 2> $O/c21_$v.err || { echo "$v failed"; tail -5 $O/c21_$v.err; cp /tmp/keep.so $P/libmpo_hip.so; exit 1; }
  python - <<PY
import json
d=json.load(open('$O/c21_$v.json')); r=d['roofline']; print('$v', d['value'], d['ms_per_step'], r.get('achieved'), r.get('cross_attention',{}).get('achieved'))
PY
done
cp /tmp/keep.so $P/libmpo_hip.so
