O=gpurun_out/r04; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_patch_wgrad.py tests/test_gpu_patch_coattn.py tests/test_gpu_models.py tests/test_gpu_graph.py tests/test_reference_swap.py -m gpu -x -q > $O/c22_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -3 $O/c22_tests.log
if [ $rc -ne 0 ]; then grep -E "^E |FAILED" $O/c22_tests.log | head -30; exit $rc; fi
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > $O/c22_bench.json 2> $O/c22_bench.err && python - <<PY
import json
d=json.load(open('$O/c22_bench.json')); print('bench', d['value'], d['ms_per_step'])
PY
