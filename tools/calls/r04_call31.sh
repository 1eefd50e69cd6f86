O=gpurun_out/r04; mkdir -p $O
for e in "X=0" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "DEBUG_HIP_GRAPH_SEGMENT_SCHEDULING=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "HIP_MEM_POOL_USE_VM=0"; do
  echo "== $e"
  env $e timeout -k 10 60 tools/_bin/flagchain 200 2>&1 | grep "mode A"
  env $e timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline > $O/c31.json 2> $O/c31.err && python - <<PY
import json
d=json.load(open('$O/c31.json')); print('   bench', d['value'], d['ms_per_step'])
PY
done
