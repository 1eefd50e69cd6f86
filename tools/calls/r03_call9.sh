# GPU call 9: PMC evidence.  (a) the headline step, eager launches (the kernels inside their workload): FETCH_SIZE / WRITE_SIZE / SQ
# passes -> f1, K1 backward (two-wave), dW_H;  (b) row f3: SQ pass over the bag self-attention kernels at 15 000 rows.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for pass in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 240 rocprofv3 --pmc $pass -d $R/gpurun_out/pmc_mcat_$pass -o p --output-format csv -- python3 $R/bench.py --no-extras --no-cpu-baseline --no-graph --steps 6 --warmup 2 > $R/gpurun_out/pmc_mcat_$pass.log 2>&1; rc=$?
  echo "pmc mcat $pass rc=$rc"; if [ $rc -ne 0 ]; then tail -5 $R/gpurun_out/pmc_mcat_$pass.log; exit $rc; fi
done
timeout -k 10 240 rocprofv3 --pmc $SQ -d $R/gpurun_out/pmc_mcat_SQ -o p --output-format csv -- python3 $R/bench.py --no-extras --no-cpu-baseline --no-graph --steps 6 --warmup 2 > $R/gpurun_out/pmc_mcat_SQ.log 2>&1; rc=$?
echo "pmc mcat SQ rc=$rc"; if [ $rc -ne 0 ]; then tail -5 $R/gpurun_out/pmc_mcat_SQ.log; exit $rc; fi
SQ2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT"
timeout -k 10 300 rocprofv3 --pmc $SQ2 -d $R/gpurun_out/pmc_ge_SQ -o p --output-format csv -- python3 $R/tools/gpu_time_ge.py 15000 3 train > $R/gpurun_out/pmc_ge_SQ.log 2>&1; rc=$?
echo "pmc ge SQ rc=$rc"; if [ $rc -ne 0 ]; then tail -5 $R/gpurun_out/pmc_ge_SQ.log; fi
cd $R
python tools/pmc_summarize.py gpurun_out/r03_pmc_mcat.json gpurun_out/pmc_mcat_FETCH_SIZE gpurun_out/pmc_mcat_WRITE_SIZE gpurun_out/pmc_mcat_SQ --match patch_coattn_fwd_kernel coattn_bwd8_kernel patch_wgrad_kernel | cut -c1-600
python tools/pmc_summarize.py gpurun_out/r03_pmc_ge.json gpurun_out/pmc_ge_SQ --match bag_sa_b3 | cut -c1-600
du -sh gpurun_out/pmc_* | tail -5
exit 0
