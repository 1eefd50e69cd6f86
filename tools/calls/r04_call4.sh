# r04 call 4: f1 v2 (direct H_bag stores, register prefetch of the next block's X, 5 + 3 ring): tests, timing, ablations, stamps
O=gpurun_out/r04; mkdir -p $O
P=multimodal_path_omic_amd
timeout -k 10 500 python -m pytest tests/test_gpu_patch_coattn.py -m gpu -x -q > $O/c4_f1_tests.log 2>&1; rc=$?
echo "f1 tests rc=$rc"; tail -5 $O/c4_f1_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
cp $P/libmpo_hip.so /tmp/keep.so
rm -f $O/c4_ablate.log
for v in keep noepi noepinomma; do
  if [ $v = keep ]; then cp /tmp/keep.so $P/libmpo_hip.so; else cp $P/libmpo_hip_$v.so $P/libmpo_hip.so; fi
  echo "== $v" >> $O/c4_ablate.log
  timeout -k 10 120 python tools/gpu_time_f1.py >> $O/c4_ablate.log 2>&1 || exit 1
done
cp $P/libmpo_hip_stamps.so $P/libmpo_hip.so
timeout -k 10 120 python tools/gpu_f1_stamps.py >> $O/c4_ablate.log 2>&1
cp /tmp/keep.so $P/libmpo_hip.so
timeout -k 10 120 python tools/gpu_time_wgrad.py >> $O/c4_ablate.log 2>&1
grep -v amdgpu.ids $O/c4_ablate.log
