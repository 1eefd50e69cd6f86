# r04 call 15: A/B of the weight-gradient kernel on 224 workgroups and of bag plans on fewer workgroups, inside the MCAT window step
O=gpurun_out/r04; mkdir -p $O
for v in "base" "wg224 --wgrad-workgroups 224" "plan240 --plan-workgroups 240" "plan224 --plan-workgroups 224 --wgrad-workgroups 224" "base2"; do
  set -- $v; n=$1; shift
  timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline "$@" > $O/c15_$n.json 2> $O/c15_$n.err || { tail -5 $O/c15_$n.err; exit 1; }
  python - <<PY
import json
d=json.load(open('$O/c15_$n.json'))
print('$n', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline'].get('cross_attention',{}).get('avg_launch_us'))
PY
done
