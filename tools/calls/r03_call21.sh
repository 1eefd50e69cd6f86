mkdir -p gpurun_out
timeout -k 10 500 python tools/gpu_time_ge_sizes.py 15000 > gpurun_out/r03_ge_sizes.txt 2>&1; echo "rc=$?"
grep -v "amdgpu.ids" gpurun_out/r03_ge_sizes.txt | tail -8
