O=gpurun_out/r04; mkdir -p $O
P=multimodal_path_omic_amd
cp $P/libmpo_hip.so /tmp/keep.so
cp $P/libmpo_hip_stamps.so $P/libmpo_hip.so
timeout -k 10 200 python tools/gpu_f1_stamps.py > $O/c25_stamps.txt 2>&1; rc=$?
cp /tmp/keep.so $P/libmpo_hip.so
cat $O/c25_stamps.txt
exit $rc
