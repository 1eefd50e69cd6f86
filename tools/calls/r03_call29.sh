# the driver's own command, five times in a row (fresh process each): how stable is a 22 ms timed region?
mkdir -p gpurun_out
for i in 1 2 3 4 5; do
  timeout -k 10 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/r03_b29_$i.json 2> gpurun_out/r03_b29_$i.err || { echo "run $i failed"; tail -3 gpurun_out/r03_b29_$i.err; exit 1; }
  python3 - "$i" <<'PY'
import json,sys
d=json.loads([l for l in open(f'gpurun_out/r03_b29_{sys.argv[1]}.json') if l.startswith('{')][-1])
print(sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])
PY
done
