O=gpurun_out/r04; mkdir -p $O
P=multimodal_path_omic_amd
cp $P/libmpo_hip.so /tmp/keep.so
rm -f $O/c10.log
for v in stamps stampsmma; do
  cp $P/libmpo_hip_$v.so $P/libmpo_hip.so
  echo "== $v" >> $O/c10.log
  timeout -k 10 120 python tools/gpu_f1_stamps.py >> $O/c10.log 2>&1 || { cp /tmp/keep.so $P/libmpo_hip.so; tail -5 $O/c10.log; exit 1; }
done
cp /tmp/keep.so $P/libmpo_hip.so
grep -v amdgpu.ids $O/c10.log
