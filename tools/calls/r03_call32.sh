# output-stationary patch-layer kernel: parity (patch tests, NaCAGaT + GE models), then the NaCAGaT bench leg + kernel stats
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 120 python -m pytest tests/test_gpu_patch_coattn.py -x -q -m gpu -k "patch_layer_alone" > gpurun_out/r03_t32a.log 2>&1; rc=$?; echo "first test rc=$rc"
tail -3 gpurun_out/r03_t32a.log
if [ $rc -ne 0 ]; then grep -E "Error|assert" gpurun_out/r03_t32a.log | head -10; exit 1; fi
timeout -k 10 900 python -m pytest tests/test_gpu_patch_coattn.py tests/test_gpu_coattn_nacagat.py tests/test_gpu_models.py tests/test_gpu_bag_selfattn.py -x -q -m gpu > gpurun_out/r03_t32.log 2>&1; echo "tests rc=$?"
tail -2 gpurun_out/r03_t32.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_b32 -o p --output-format csv -- python3 $R/bench.py --model nacagat --steps 30 --warmup 5 --no-extras --no-cpu-baseline > $R/gpurun_out/r03_b32.json 2> $R/gpurun_out/r03_b32.err; echo "rocprof rc=$?"
cd $R
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03_b32.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'])
PY
grep "patch_fc_bf16\|patch_coattn_fwd\|pack_patch" gpurun_out/prof_b32/p_kernel_stats.csv | sed 's/"[^"]*",/K,/' | cut -c1-90
