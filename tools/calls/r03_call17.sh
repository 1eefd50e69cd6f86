mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_patch_coattn.py tests/test_gpu_models.py tests/test_gpu_graph.py tests/test_gpu_cohort.py tests/test_gpu_coattn_nacagat.py -m gpu -q > gpurun_out/r03_t17.log 2>&1; rc=$?
echo "tests rc=$rc"; grep -E "passed|failed" gpurun_out/r03_t17.log | tail -2
if [ $rc -ne 0 ]; then grep -E "^FAILED|Error" gpurun_out/r03_t17.log | head; exit $rc; fi
for i in 1 2; do timeout -k 10 120 python bench.py --no-extras --no-cpu-baseline > gpurun_out/r03_b17_$i.json 2>gpurun_out/r03_b17.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r03_b17_$i.json')); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"; done
timeout -k 10 120 python bench.py --no-extras --no-cpu-baseline --model nacagat > gpurun_out/r03_b17_n.json 2>gpurun_out/r03_b17.err; python -c "
import json; d=json.load(open('gpurun_out/r03_b17_n.json')); print('nacagat', d['value'], d['ms_per_step'])"
