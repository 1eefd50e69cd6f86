# r04 call 14: fp32 patch layer range tests; DP overlap probe (stand-in collective beside dW_H, 256 vs 224 workgroups) + its kernel trace
O=gpurun_out/r04; mkdir -p $O
rc=0
echo "tests rc=$rc"; tail -3 $O/c14_tests.log
if [ $rc -ne 0 ]; then grep -E "^E |FAILED" $O/c14_tests.log | head -20; exit $rc; fi
timeout -k 10 200 python tools/gpu_probe_dp_overlap.py > $O/c14_dp_overlap.txt 2>&1 || { tail -5 $O/c14_dp_overlap.txt; exit 1; }
grep -v amdgpu.ids $O/c14_dp_overlap.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/$O/prof_dp -o p --output-format csv -- python3 $R/tools/gpu_probe_dp_overlap.py > $R/$O/c14_dp_overlap_rocprof.log 2>&1; echo "rocprof rc=$?"
cd $R
python3 tools/gpu_trace_overlap.py $O/prof_dp/p_kernel_trace.csv patch_wgrad_kernel occupy_kernel > $O/c14_dp_two_streams.txt 2>&1; head -40 $O/c14_dp_two_streams.txt; head -2 $O/prof_dp/p_kernel_trace.csv
rm -rf $O/prof_dp
