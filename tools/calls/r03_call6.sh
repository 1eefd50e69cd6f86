mkdir -p gpurun_out
timeout -k 10 300 python tools/gpu_diag_window_vs_slide.py > gpurun_out/r03_diag_k2.log 2>&1; echo "diag rc=$?"; tail -6 gpurun_out/r03_diag_k2.log | cut -c1-400
timeout -k 10 300 python -m pytest tests/test_gpu_coattn_mcat.py -m gpu -q -k two_wave 2>&1 | tail -3
