# Round 4 final, part 2 (part 1 = r04_final_tests.sh): the bench line (with extras and CPU baselines), rocprofv3 kernel stats +
# step timelines of the bench command for the headline / NaCAGaT / ragged / 100k-fp32 configurations and the f3 step.
# Outputs under gpurun_out/final/ (copy what is to be judged into profiles/).
mkdir -p gpurun_out/final
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final

timeout -k 10 400 python bench.py > $O/bench_mcat_n1.json 2> $O/bench_mcat_n1.err; rc=$?
echo "bench rc=$rc"; if [ $rc -ne 0 ]; then tail -5 $O/bench_mcat_n1.err; exit $rc; fi
python - <<'PY'
import json
d=json.load(open('gpurun_out/final/bench_mcat_n1.json'))
def show(n,r):
    rf=r.get('roofline') or {}
    print(n, r.get('value'), r.get('ms_per_step'), rf.get('avg_launch_us'), rf.get('frac'), r.get('error'), (r.get('cpu_baseline') or {}).get('value'))
show('headline',d)
for k,v in d.get('extra',{}).items(): show(k,v)
PY
cd /tmp && export TMPDIR=/tmp
run_prof() {   # name, bench args...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o p --output-format csv -- python3 $R/bench.py --no-extras --no-cpu-baseline "$@" > $O/bench_${name}_under_rocprof.json 2> $O/bench_${name}_under_rocprof.err; rc=$?
  echo "rocprof $name rc=$rc"
  if [ $rc -ne 0 ]; then return $rc; fi
  python3 $R/tools/gpu_trace_step.py $O/prof_$name/p_kernel_trace.csv > $O/${name}_step_timeline.txt 2>&1
  head -1 $O/${name}_step_timeline.txt
  cp $O/prof_$name/p_kernel_stats.csv $O/${name}_kernel_stats.csv
  rm -f $O/prof_$name/p_kernel_trace.csv
}
run_prof mcat || exit 1
run_prof nacagat --model nacagat || exit 1
run_prof ragged --ragged || exit 1
run_prof mcat_f32_100k --patches 100000 --dtype f32 --window 8 --steps 8 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_ge -o p --output-format csv -- python3 $R/tools/gpu_time_ge.py 15000 5 train > $O/ge_under_rocprof.log 2>&1; echo "rocprof ge rc=$?"
cp $O/prof_ge/p_kernel_stats.csv $O/ge_kernel_stats.csv
rm -f $O/prof_ge/p_kernel_trace.csv
cd $R
du -sh gpurun_out/final
