O=gpurun_out/r04; mkdir -p $O
P=multimodal_path_omic_amd
timeout -k 10 600 python -m pytest tests/test_gpu_patch_wgrad.py tests/test_gpu_patch_coattn.py tests/test_gpu_coattn_mcat.py -m gpu -x -q > $O/c19_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -3 $O/c19_tests.log
if [ $rc -ne 0 ]; then grep -E "^E |FAILED" $O/c19_tests.log | head -20; exit $rc; fi
bash tools/calls/r04_prof.sh mcat_wgback | grep -E "rocprof|step span|patch_wgrad_kernel|coattn_bwd8" || exit 1
cp $P/libmpo_hip.so /tmp/keep.so; cp $P/libmpo_hip_wgfwd.so $P/libmpo_hip.so
bash tools/calls/r04_prof.sh mcat_wgfwd | grep -E "rocprof|step span|patch_wgrad_kernel|coattn_bwd8"; rc=$?
cp /tmp/keep.so $P/libmpo_hip.so
bash tools/calls/r04_prof.sh mcat_wgback2 | grep -E "rocprof|step span|patch_wgrad_kernel|coattn_bwd8"
exit $rc
