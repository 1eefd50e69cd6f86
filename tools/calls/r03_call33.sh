# kernel trace of the two-stream overlap probe (product work plan only): which launches of the other stream run under a bag kernel
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_overlap -o p --output-format csv -- python3 $R/tools/gpu_probe_overlap.py 16 15000 none > $R/gpurun_out/r03_overlap_traced.log 2>&1; echo "rc=$?"
cd $R
grep "plan workgroups" gpurun_out/r03_overlap_traced.log
f=$(ls gpurun_out/prof_overlap/*kernel_trace.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then python tools/gpu_trace_two_streams.py "$f" > gpurun_out/r03_overlap_two_streams.txt 2>&1; cat gpurun_out/r03_overlap_two_streams.txt | cut -c1-170; rm -f "$f"; else echo "no trace"; tail -5 gpurun_out/r03_overlap_traced.log; fi
