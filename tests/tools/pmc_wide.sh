# HBM traffic of the Cin = 256 conv kernel (conv4d_kernel) inside an EDSR 256 x 32 training step, from PMC counters: separate rocprofv3 --pmc
# passes (MI355X_MICROARCH.md: one counter group per pass, --kernel-trace only) over a short bench run.   usage (GPU box): bash tests/tools/pmc_wide.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmcw_$c -o p --output-format csv -- python3 $R/bench.py --model edsr256 --batch 16 --steps 2 --warmup 1 --probe-steps 1 --no-cpu-baseline > /dev/null 2>&1
done
cd $R
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob('gpurun_out/pmcw_*')):
    for f in glob.glob(d+'/*counter_collection.csv'):
        by=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'conv4d_kernel' in r['Kernel_Name'] or 'wgrad_dma_kernel<4>' in r['Kernel_Name']:
                by[(r['Kernel_Name'][:40], r['Counter_Name'], r.get('Grid_Size','') or r.get('Grid_Size_X',''))].append(float(r['Counter_Value']))
        for k,v in sorted(by.items()):
            v=v[len(v)//2:]
            print(d.split('/')[-1], k, 'mean per launch %.1f KB' % (sum(v)/len(v)), 'n', len(v))
PY
