O=gpurun_out/r04; mkdir -p $O
P=multimodal_path_omic_amd
bash tools/calls/r04_prof.sh mcat_backwalk || exit 1
cp $P/libmpo_hip.so /tmp/keep.so; cp $P/libmpo_hip_k1fwd.so $P/libmpo_hip.so
bash tools/calls/r04_prof.sh mcat_fwdwalk; rc=$?
cp /tmp/keep.so $P/libmpo_hip.so
exit $rc
