# kernel stats of the NaCAGaT step (one-pass K2 key gradient)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_b23 -o p --output-format csv -- python3 $R/bench.py --model nacagat --steps 30 --warmup 5 --no-extras --no-cpu-baseline > $R/gpurun_out/r03_b23_prof.json 2> $R/gpurun_out/r03_b23_prof.err; echo "rocprof rc=$?"
cd $R
f=$(ls gpurun_out/prof_b23/*/*kernel_stats.csv gpurun_out/prof_b23/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then head -16 "$f" | cut -c1-170; else echo "no stats file"; fi
