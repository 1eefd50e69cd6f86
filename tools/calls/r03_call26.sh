# K1 backward for fp32 bags on the vector ALUs: parity (K1 + model tests), then cfg 5's bench leg and kernel stats
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_coattn_mcat.py tests/test_gpu_models.py tests/test_gpu_graph.py -x -q -m gpu -s > gpurun_out/r03_t26.log 2>&1; echo "tests rc=$?"
grep -E "K1 fp32 backward|passed|failed|FAILED" gpurun_out/r03_t26.log | tail -8
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_b26 -o p --output-format csv -- python3 $R/bench.py --no-extras --no-cpu-baseline --patches 100000 --dtype f32 --window 8 --steps 8 > $R/gpurun_out/r03_b26_prof.json 2> $R/gpurun_out/r03_b26_prof.err; echo "rocprof rc=$?"
cd $R
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03_b26_prof.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'])
PY
f=$(ls gpurun_out/prof_b26/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then grep "coattn_bwd\|patch_fc_f32\|wgrad_f32_kernel\|fwd_partial" "$f" | sed 's/"[^"]*",/K,/' | cut -c1-100; else echo "no stats file"; fi
