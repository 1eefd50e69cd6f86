O=gpurun_out/r04; mkdir -p $O
P=multimodal_path_omic_amd
cp $P/libmpo_hip.so /tmp/keep.so
for v in keep xnt dma dmant xonly wonly mmaonly; do
  if [ $v = keep ]; then cp /tmp/keep.so $P/libmpo_hip.so; else cp $P/libmpo_hip_$v.so $P/libmpo_hip.so; fi
  echo "== $v" >> $O/c8.log
  timeout -k 10 120 python tools/gpu_time_f1.py >> $O/c8.log 2>&1 || exit 1
done
cp /tmp/keep.so $P/libmpo_hip.so
grep -v amdgpu.ids $O/c8.log
