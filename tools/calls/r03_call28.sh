# fp32 weight gradient with the column blocks of a row range on one XCD: parity, kernel stats of cfg 5, FETCH_SIZE pass
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_patch_fc_f32.py tests/test_gpu_models.py -x -q -m gpu > gpurun_out/r03_t28.log 2>&1; echo "tests rc=$?"
tail -2 gpurun_out/r03_t28.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_b28 -o p --output-format csv -- python3 $R/bench.py --no-extras --no-cpu-baseline --patches 100000 --dtype f32 --window 8 --steps 8 > $R/gpurun_out/r03_b28_prof.json 2> $R/gpurun_out/r03_b28_prof.err; echo "rocprof rc=$?"
rm -rf $R/gpurun_out/pmc_f32_FETCH_SIZE
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_f32_FETCH_SIZE -o p --output-format csv -- python3 $R/bench.py --no-extras --no-cpu-baseline --no-graph --patches 100000 --dtype f32 --window 8 --steps 4 --warmup 1 > $R/gpurun_out/pmc_f32_FETCH_SIZE.log 2>&1; echo "pmc rc=$?"
cd $R
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03_b28_prof.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'])
PY
f=$(ls gpurun_out/prof_b28/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then grep "coattn_bwd\|patch_fc_f32\|wgrad_f32_kernel\|fwd_partial" "$f" | sed 's/"[^"]*",/K,/' | cut -c1-100; fi
python tools/pmc_summarize.py gpurun_out/r03_pmc_f32_fetch2.json gpurun_out/pmc_f32_FETCH_SIZE --match patch_wgrad_f32_kernel | cut -c1-300
