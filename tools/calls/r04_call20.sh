O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_tail.py tests/test_gpu_models.py tests/test_gpu_graph.py -m gpu -x -q > $O/c20_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -3 $O/c20_tests.log
if [ $rc -ne 0 ]; then grep -E "^E |FAILED" $O/c20_tests.log | head -20; exit $rc; fi
bash tools/calls/r04_prof.sh mcat_poolfused | head -6
grep -n "pool_score\|pool_fwd\|pool_bwd" $O/mcat_poolfused_step_timeline.txt | cut -c1-110
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > $O/c20_bench.json 2> $O/c20_bench.err && python - <<PY
import json
d=json.load(open('$O/c20_bench.json')); print('bench', d['value'], d['ms_per_step'])
PY
