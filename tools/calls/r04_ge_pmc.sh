# SQ counters of the row-f3 attention kernels inside the GE training step (separate --pmc pass, no trace domains)
O=gpurun_out/r04; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for pass in A B; do
  if [ $pass = A ]; then ctr="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; else ctr="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_MFMA"; fi
  timeout -k 10 300 rocprofv3 --pmc $ctr -d $R/$O/gepmc_$pass -o p --output-format csv -- python3 $R/tools/gpu_time_ge.py 15000 3 train > $R/$O/gepmc_$pass.log 2>&1; rc=$?
  echo "pmc $pass rc=$rc"; if [ $rc -ne 0 ]; then tail -5 $R/$O/gepmc_$pass.log; exit $rc; fi
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for pas in 'AB':
    f=glob.glob(f'gpurun_out/r04/gepmc_{pas}/**/*counter_collection.csv', recursive=True)
    if not f: print('no csv', pas); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k=r['Kernel_Name']
        if 'bag_sa_b3' not in k: continue
        k=k[k.index('bag_sa_b3'):][:28]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in agg.items():
        print(pas, k, {a: f'{b:.3g}' for a,b in v.items()})
PY
rm -rf $O/gepmc_A $O/gepmc_B
