# after the K2 key-gradient change: the whole GPU suite + smoke
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r03_t25.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/r03_t25.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03_smoke25.log 2>&1; echo "smoke rc=$?"
tail -3 gpurun_out/r03_smoke25.log
