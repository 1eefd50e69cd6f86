# gene-expression model 'big' (one head of 512): attention core, dropout mask, whole model
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_bag_selfattn.py -x -q -m gpu > gpurun_out/r03_t20.log 2>&1; echo "tests rc=$?"
tail -15 gpurun_out/r03_t20.log
