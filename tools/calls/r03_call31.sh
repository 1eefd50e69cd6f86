# 16-row steps dealt to the waves in the two vector kernels: parity, then kernel stats of the NaCAGaT and 100k-fp32 steps
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_coattn_nacagat.py tests/test_gpu_coattn_mcat.py tests/test_gpu_models.py -x -q -m gpu > gpurun_out/r03_t31.log 2>&1; echo "tests rc=$?"
tail -2 gpurun_out/r03_t31.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_b31n -o p --output-format csv -- python3 $R/bench.py --model nacagat --steps 30 --warmup 5 --no-extras --no-cpu-baseline > $R/gpurun_out/r03_b31n.json 2> $R/gpurun_out/r03_b31n.err; echo "rocprof nacagat rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_b31f -o p --output-format csv -- python3 $R/bench.py --no-extras --no-cpu-baseline --patches 100000 --dtype f32 --window 8 --steps 8 > $R/gpurun_out/r03_b31f.json 2> $R/gpurun_out/r03_b31f.err; echo "rocprof f32 rc=$?"
cd $R
for t in n f; do
python - "$t" <<'PY'
import json,sys
d=json.loads([l for l in open(f'gpurun_out/r03_b31{sys.argv[1]}.json') if l.startswith('{')][-1])
print(sys.argv[1], d['value'], d['ms_per_step'])
PY
done
grep "key_grad" gpurun_out/prof_b31n/p_kernel_stats.csv | sed 's/"[^"]*",/K,/' | cut -c1-80
grep "coattn_bwd_f32" gpurun_out/prof_b31f/p_kernel_stats.csv | sed 's/"[^"]*",/K,/' | cut -c1-80
