# HBM traffic of the one-launch RCAB kernels from PMC counters: separate rocprofv3 --pmc passes over a short RCAN bench run
# (MI355X_MICROARCH.md: one counter group per pass, --kernel-trace only).   usage (GPU box): bash tools/pmc_rcab.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmcr_$c -o p --output-format csv -- python3 $R/bench.py --model rcan --steps 2 --warmup 1 --probe-steps 1 --no-cpu-baseline > /dev/null 2>&1
done
cd $R
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob('gpurun_out/pmcr_*')):
    for f in glob.glob(d+'/*counter_collection.csv'):
        by=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'rcab_kernel' in r['Kernel_Name']:
                by[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k,v in sorted(by.items()):
            v=v[len(v)//2:]
            print(d, k, 'mean per launch %.1f' % (sum(v)/len(v)), 'n', len(v))
PY
