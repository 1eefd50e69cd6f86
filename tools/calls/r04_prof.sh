# rocprofv3 kernel stats + step timeline of the bench command for one configuration: bash tools/calls/r04_prof.sh NAME [bench args]
O=gpurun_out/r04; mkdir -p $O
R=$GRAFT_REPO_ROOT
name=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$name -o p --output-format csv -- python3 $R/bench.py --no-extras --no-cpu-baseline "$@" > $R/$O/bench_${name}_under_rocprof.json 2> $R/$O/bench_${name}_under_rocprof.err; rc=$?
echo "rocprof $name rc=$rc"
cd $R
if [ $rc -ne 0 ]; then tail -5 $O/bench_${name}_under_rocprof.err; exit $rc; fi
python3 tools/gpu_trace_step.py $O/prof_$name/p_kernel_trace.csv > $O/${name}_step_timeline.txt 2>&1
head -1 $O/${name}_step_timeline.txt
cp $O/prof_$name/p_kernel_stats.csv $O/${name}_kernel_stats.csv
rm -f $O/prof_$name/p_kernel_trace.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$O/${name}_kernel_stats.csv')))
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
PY
