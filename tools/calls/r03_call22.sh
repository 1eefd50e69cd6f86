# K2 backward with the query-side accumulations folded into the bag-side pass: parity, then the NaCAGaT bench leg + kernel stats
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_coattn_nacagat.py tests/test_gpu_models.py -x -q -m gpu > gpurun_out/r03_t22.log 2>&1; echo "tests rc=$?"
tail -6 gpurun_out/r03_t22.log
timeout -k 10 300 python bench.py --model nacagat --steps 50 --warmup 10 --no-extras --no-cpu-baseline > gpurun_out/r03_b22_nacagat.json 2> gpurun_out/r03_b22_nacagat.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03_b22_nacagat.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['roofline'])
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_b22 -o b22 -- python3 $GRAFT_REPO_ROOT/bench.py --model nacagat --steps 30 --warmup 5 --no-extras --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r03_b22_prof.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/prof_b22/*/*kernel_stats.csv gpurun_out/prof_b22/*kernel_stats.csv 2>/dev/null | head -1); echo $f; head -14 $f | cut -c1-150
