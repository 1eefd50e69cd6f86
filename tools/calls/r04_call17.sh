# r04 call 17: the weight gradient beside the rest of the backward pass (second stream inside the captured graph): graph tests, A/B in the bench
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_graph.py tests/test_gpu_dp.py tests/test_gpu_tail.py -m gpu -x -q > $O/c17_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -3 $O/c17_tests.log
if [ $rc -ne 0 ]; then grep -E "^E |FAILED" $O/c17_tests.log | head -20; exit $rc; fi
for v in "side" "main --no-side-wgrad" "side2" "main2 --no-side-wgrad" "nacagat_side --model nacagat" "nacagat_main --model nacagat --no-side-wgrad"; do
  set -- $v; n=$1; shift
  timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline "$@" > $O/c17_$n.json 2> $O/c17_$n.err || { tail -5 $O/c17_$n.err; exit 1; }
  python - <<PY
import json
d=json.load(open('$O/c17_$n.json'))
print('$n', d['value'], d['ms_per_step'], d['config']['launch'])
PY
done
