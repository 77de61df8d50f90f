cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in fwd bwd; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmcb_${mode}_f -o p --output-format csv -- python3 $R/tests/tools/pmc_block.py $mode > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmcb_${mode}_w -o p --output-format csv -- python3 $R/tests/tools/pmc_block.py $mode > /dev/null 2>&1
done
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $R/gpurun_out/pmcb_fwd_h -o p --output-format csv -- python3 $R/tests/tools/pmc_block.py fwd > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $R/gpurun_out/pmcb_fwd_s -o p --output-format csv -- python3 $R/tests/tools/pmc_block.py fwd > /dev/null 2>&1
cd $R
ls gpurun_out/pmcb_fwd_f
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob('gpurun_out/pmcb_*')):
    for f in glob.glob(d+'/*counter_collection.csv'):
        rows=[r for r in csv.DictReader(open(f)) if 'conv_block' in r['Kernel_Name']]
        by=collections.defaultdict(list)
        for r in rows: by[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in by.items():
            v=v[5:]
            print(d, k, 'mean per launch %.1f' % (sum(v)/len(v)), 'n', len(v))
PY
