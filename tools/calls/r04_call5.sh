# r04 call 5: the MCAT bench line (no extras) with the product library and with the r03 kernel (variant "old"), same box
O=gpurun_out/r04; mkdir -p $O
P=multimodal_path_omic_amd
cp $P/libmpo_hip.so /tmp/keep.so
for v in keep old keep old; do
  if [ $v = keep ]; then cp /tmp/keep.so $P/libmpo_hip.so; else cp $P/libmpo_hip_$v.so $P/libmpo_hip.so; fi
  timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > $O/c5_bench_$v.json 2> $O/c5_bench_$v.err || { tail -5 $O/c5_bench_$v.err; cp /tmp/keep.so $P/libmpo_hip.so; exit 1; }
  python - <<PY
import json
d=json.load(open('$O/c5_bench_$v.json'))
print('$v', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])
PY
done
cp /tmp/keep.so $P/libmpo_hip.so
