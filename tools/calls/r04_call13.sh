# r04 call 13: whole GPU suite + the whole bench line with the two-stream patch-layer kernel + K1 forward as its own launch
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $O/c13_gpu_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; grep -E "passed|failed" $O/c13_gpu_tests.log | tail -2
if [ $rc -ne 0 ]; then grep -E "^FAILED|Error|assert " $O/c13_gpu_tests.log | head -20; exit $rc; fi
timeout -k 10 400 python bench.py > $O/c13_bench.json 2> $O/c13_bench.err; rc=$?
echo "bench rc=$rc"; if [ $rc -ne 0 ]; then tail -5 $O/c13_bench.err; exit $rc; fi
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04/c13_bench.json'))
def show(n,r):
    rf=r.get('roofline') or {}
    print(n, r.get('value'), r.get('ms_per_step'), rf.get('kernel'), rf.get('avg_launch_us'), rf.get('frac'), (rf.get('cross_attention') or {}).get('avg_launch_us'), (rf.get('cross_attention') or {}).get('frac'), r.get('error'), (r.get('cpu_baseline') or {}).get('value'))
show('headline',d)
for k,v in d.get('extra',{}).items(): show(k,v)
PY
