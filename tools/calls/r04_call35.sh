O=gpurun_out/r04; mkdir -p $O
P=multimodal_path_omic_amd
cp $P/libmpo_hip.so /tmp/keep.so
for v in base vf2 base vf2 base vf2; do
  cp $P/libmpo_hip_$v.so $P/libmpo_hip.so
  for cfg in "mcat" "nacagat --model nacagat"; do
    set -- $cfg; name=$1; shift
    timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline "$@" > $O/c35_${v}_$name.json 2> $O/c35_${v}_$name.err || { echo "$v $name failed"; tail -5 $O/c35_${v}_$name.err; cp /tmp/keep.so $P/libmpo_hip.so; exit 1; }
    python - <<PY
import json
d=json.load(open('$O/c35_${v}_$name.json')); r=d['roofline']; print('$v $name', d['value'], d['ms_per_step'], r.get('avg_launch_us'), (r.get('cross_attention') or {}).get('avg_launch_us'))
PY
  done
done
cp /tmp/keep.so $P/libmpo_hip.so
