mkdir -p gpurun_out
timeout -k 10 300 python tools/gpu_time_sa_dropout.py 15000 > gpurun_out/r03_sa_dropout.txt 2>&1; echo "rc=$?"
grep -v amdgpu.ids gpurun_out/r03_sa_dropout.txt | tail -6
