# GE-NaCAGaT (row f3): parity tests, then rocprofv3 kernel stats of the training step.  bash tools/calls/r04_ge.sh NAME
O=gpurun_out/r04; mkdir -p $O
R=$GRAFT_REPO_ROOT
name=${1:-ge}
timeout -k 10 600 python -m pytest tests/test_gpu_bag_selfattn.py -m gpu -x -q > $O/${name}_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -2 $O/${name}_tests.log
if [ $rc -ne 0 ]; then grep -E "^E |FAILED" $O/${name}_tests.log | head -20; exit $rc; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$name -o p --output-format csv -- python3 $R/tools/gpu_time_ge.py 15000 5 train > $R/$O/${name}_under_rocprof.log 2>&1; rc=$?
cd $R
echo "rocprof rc=$rc"; tail -3 $O/${name}_under_rocprof.log
cp $O/prof_$name/p_kernel_stats.csv $O/${name}_kernel_stats.csv; rm -f $O/prof_$name/p_kernel_trace.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$O/${name}_kernel_stats.csv')))
for r in rows[:9]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
PY
