# the N > 1 code path of bench.py rehearsed with 2 ranks sharing the one card over gloo (RCCL refuses two ranks on one device)
mkdir -p gpurun_out
MPO_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r03_bench_n2_gloo.json 2> gpurun_out/r03_bench_n2_gloo.err; echo "bench n2 rc=$?"
tail -3 gpurun_out/r03_bench_n2_gloo.err | cut -c1-200
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r03_bench_n2_gloo.json") if l.startswith("{")][-1])
print(d['n_gpus'], d['value'], d['ms_per_step'], d['config']['launch'][:90])
for k,v in d.get('extra',{}).items(): print(k, v.get('value'), v.get('ms_per_step'), v.get('error'))
PY
