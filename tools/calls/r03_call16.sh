mkdir -p gpurun_out/final
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final
timeout -k 10 120 python tools/gpu_probe_copybw.py > $O/copybw.txt 2>&1; echo "copy rc=$?"; tail -1 $O/copybw.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_mcat_f32_100k -o p --output-format csv -- python3 $R/bench.py --no-extras --no-cpu-baseline --patches 100000 --dtype f32 --window 8 --steps 8 > $O/bench_mcat_f32_100k_under_rocprof.json 2> $O/bench_mcat_f32_100k.err; echo "rocprof rc=$?"
python3 $R/tools/gpu_trace_step.py $O/prof_mcat_f32_100k/p_kernel_trace.csv > $O/mcat_f32_100k_step_timeline.txt 2>&1
rm -f $O/prof_mcat_f32_100k/p_kernel_trace.csv
cd $R
head -1 $O/mcat_f32_100k_step_timeline.txt; cut -c1-160 $O/bench_mcat_f32_100k_under_rocprof.json
