# r04 call 2: ablations of the rebuilt f1 kernel on one box (variant libraries swapped in for the stand-alone timing)
O=gpurun_out/r04; mkdir -p $O
P=multimodal_path_omic_amd
cp $P/libmpo_hip.so /tmp/keep.so
for v in keep old nomma nodma xfull noepi noepixfull noepinomma; do
  if [ $v = keep ]; then cp /tmp/keep.so $P/libmpo_hip.so; else cp $P/libmpo_hip_$v.so $P/libmpo_hip.so; fi
  echo "== $v" >> $O/c2_ablate.log
  timeout -k 10 120 python tools/gpu_time_f1.py >> $O/c2_ablate.log 2>&1 || exit 1
done
cp /tmp/keep.so $P/libmpo_hip.so
timeout -k 10 120 python tools/gpu_time_wgrad.py >> $O/c2_ablate.log 2>&1
grep -v amdgpu.ids $O/c2_ablate.log
