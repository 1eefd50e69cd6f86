# PMC evidence for the kernels added in the second half of round 3, inside their workloads (eager launches):
#   NaCAGaT step -> bag_key_grad_kernel (traffic, VALU / MFMA busy);  100k-fp32 step -> coattn_bwd_f32_kernel, patch_fc_f32 kernels
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT"
run() {  # tag, pass name, counters..., then -- bench args
  tag=$1; pass=$2; shift 2
  ctr=""; while [ "$1" != "--" ]; do ctr="$ctr $1"; shift; done; shift
  rm -rf $R/gpurun_out/pmc_${tag}_$pass
  timeout -k 10 300 rocprofv3 --pmc $ctr -d $R/gpurun_out/pmc_${tag}_$pass -o p --output-format csv -- python3 $R/bench.py --no-extras --no-cpu-baseline --no-graph "$@" > $R/gpurun_out/pmc_${tag}_$pass.log 2>&1; rc=$?
  echo "pmc $tag $pass rc=$rc"; if [ $rc -ne 0 ]; then tail -5 $R/gpurun_out/pmc_${tag}_$pass.log; return $rc; fi
}
run nacagat FETCH_SIZE FETCH_SIZE -- --model nacagat --steps 6 --warmup 2 || exit 1
run nacagat WRITE_SIZE WRITE_SIZE -- --model nacagat --steps 6 --warmup 2 || exit 1
run nacagat SQ $SQ -- --model nacagat --steps 6 --warmup 2 || exit 1
run f32 FETCH_SIZE FETCH_SIZE -- --patches 100000 --dtype f32 --window 8 --steps 4 --warmup 1 || exit 1
run f32 WRITE_SIZE WRITE_SIZE -- --patches 100000 --dtype f32 --window 8 --steps 4 --warmup 1 || exit 1
run f32 SQ $SQ -- --patches 100000 --dtype f32 --window 8 --steps 4 --warmup 1 || exit 1
cd $R
python tools/pmc_summarize.py gpurun_out/r03_pmc_nacagat.json gpurun_out/pmc_nacagat_FETCH_SIZE gpurun_out/pmc_nacagat_WRITE_SIZE gpurun_out/pmc_nacagat_SQ --match bag_key_grad_kernel k2_patch_grad_kernel key_proj_kernel bag_rowdot_gated_exact | cut -c1-900
python tools/pmc_summarize.py gpurun_out/r03_pmc_mcat_f32_100k.json gpurun_out/pmc_f32_FETCH_SIZE gpurun_out/pmc_f32_WRITE_SIZE gpurun_out/pmc_f32_SQ --match coattn_bwd_f32_kernel patch_fc_f32_kernel patch_wgrad_f32_kernel coattn_fwd_partial | cut -c1-900
find gpurun_out/pmc_nacagat_* gpurun_out/pmc_f32_* -name "*.csv" -size +2M -delete
du -sh gpurun_out/pmc_* | tail -8
exit 0
