# K2 backward: the one-pass key gradient as a VALU kernel (bag_key_grad_kernel): parity, bench leg, kernel stats
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_coattn_nacagat.py tests/test_gpu_models.py -x -q -m gpu > gpurun_out/r03_t24.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/r03_t24.log
timeout -k 10 300 python bench.py --model nacagat --steps 50 --warmup 10 --no-extras --no-cpu-baseline > gpurun_out/r03_b24_nacagat.json 2> gpurun_out/r03_b24_nacagat.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03_b24_nacagat.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'])
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_b24 -o p --output-format csv -- python3 $R/bench.py --model nacagat --steps 30 --warmup 5 --no-extras --no-cpu-baseline > $R/gpurun_out/r03_b24_prof.json 2> $R/gpurun_out/r03_b24_prof.err; echo "rocprof rc=$?"
cd $R
f=$(ls gpurun_out/prof_b24/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then grep "key_grad\|outer_gated\|colacc_gated" "$f" | sed 's/"[^"]*",/K,/' | cut -c1-100; else echo "no stats file"; fi
