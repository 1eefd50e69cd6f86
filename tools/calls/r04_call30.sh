O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_coattn_nacagat.py tests/test_gpu_tail.py tests/test_gpu_models.py tests/test_gpu_graph.py tests/test_gpu_cohort.py tests/test_reference_swap.py -m gpu -x -q > $O/c30_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -2 $O/c30_tests.log
if [ $rc -ne 0 ]; then grep -E "^E |FAILED" $O/c30_tests.log | head -20; exit $rc; fi
bash tools/calls/r04_prof.sh nacagat_noglue --model nacagat | head -4
grep -c "at::native" $O/nacagat_noglue_step_timeline.txt
grep "at::native" $O/nacagat_noglue_step_timeline.txt | cut -c1-120
timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --model nacagat > $O/c30_bench.json 2> $O/c30_bench.err && python - <<PY
import json
d=json.load(open('$O/c30_bench.json')); print('bench nacagat', d['value'], d['ms_per_step'])
PY
