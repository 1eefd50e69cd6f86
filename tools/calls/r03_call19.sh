# cost of a grid-wide barrier inside one persistent kernel (tools/probes/gridbarrier.hip, built here into tools/_bin/)
mkdir -p gpurun_out
timeout -k 10 120 tools/_bin/gridbarrier > gpurun_out/r03_gridbarrier.txt 2>&1; echo "rc=$?"
cat gpurun_out/r03_gridbarrier.txt
