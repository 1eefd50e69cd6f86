O=gpurun_out/r04; mkdir -p $O
for size in small big; do for model in mcat nacagat; do
  bash tools/calls/r04_prof.sh ${model}_${size} --model $model --model-size $size --steps 5 > $O/c23_${model}_${size}.txt 2>&1 || { cat $O/c23_${model}_${size}.txt | tail -20; exit 1; }
  head -12 $O/c23_${model}_${size}.txt
  echo "library GEMM kernels (Cijk_) in ${model} ${size}: $(grep -c Cijk_ $O/${model}_${size}_kernel_stats.csv)"
  python - <<PY
import json
d=json.loads(open('$O/bench_${model}_${size}_under_rocprof.json').read().strip().splitlines()[-1]); print('$model $size', d['value'], d['ms_per_step'])
PY
done; done
