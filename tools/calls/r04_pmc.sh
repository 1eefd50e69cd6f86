# r04 PMC evidence: FETCH_SIZE / WRITE_SIZE / SQ passes (separate runs, --pmc only) over the eager window step of
# (a) MCAT bf16 15k (patch_fc_fwd, K1 forward / backward, dW_H), (b) NaCAGaT bf16 15k (K2 score pass, key projection, patch layer),
# (c) MCAT fp32 100k (K1 fp32 forward, fp32 patch layer, fp32 weight gradient): the kernels INSIDE their workloads.
mkdir -p gpurun_out/r04
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
cd /tmp && export TMPDIR=/tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run() {   # tag, bench args
  tag=$1; shift
  for pass in FETCH_SIZE WRITE_SIZE SQ; do
    if [ $pass = SQ ]; then ctr="$SQ"; else ctr=$pass; fi
    timeout -k 10 300 rocprofv3 --pmc $ctr -d $O/pmc_${tag}_$pass -o p --output-format csv -- python3 $R/bench.py --no-extras --no-cpu-baseline --no-graph --steps 6 --warmup 2 --settle 2 "$@" > $O/pmc_${tag}_$pass.log 2>&1; rc=$?
    echo "pmc $tag $pass rc=$rc"; if [ $rc -ne 0 ]; then tail -5 $O/pmc_${tag}_$pass.log; return $rc; fi
  done
}
run mcat || exit 1
run nacagat --model nacagat || exit 1
run f32 --patches 100000 --dtype f32 --window 8 || exit 1
cd $R
python tools/pmc_summarize.py $O/r04_pmc_mcat.json $O/pmc_mcat_FETCH_SIZE $O/pmc_mcat_WRITE_SIZE $O/pmc_mcat_SQ --match patch_fc_fwd_kernel coattn_fwd_partial_kernel coattn_bwd8_kernel patch_wgrad_kernel | cut -c1-300
python tools/pmc_summarize.py $O/r04_pmc_nacagat.json $O/pmc_nacagat_FETCH_SIZE $O/pmc_nacagat_WRITE_SIZE $O/pmc_nacagat_SQ --match patch_fc_fwd_kernel key_proj_kernel bag_rowdot_gated_exact bag_key_grad_kernel k2_patch_grad_kernel patch_wgrad_kernel | cut -c1-300
python tools/pmc_summarize.py $O/r04_pmc_mcat_f32_100k.json $O/pmc_f32_FETCH_SIZE $O/pmc_f32_WRITE_SIZE $O/pmc_f32_SQ --match patch_fc_f32_kernel coattn_fwd_partial_kernel coattn_bwd_f32_kernel patch_wgrad_f32_kernel | cut -c1-300
rm -rf $O/pmc_*_FETCH_SIZE $O/pmc_*_WRITE_SIZE $O/pmc_*_SQ
ls -la $O/r04_pmc_*.json
