# HBM traffic of the block kernels (conv_block_kernel / rcab_kernel) from PMC counters, measured INSIDE the training step: separate
# rocprofv3 --pmc passes (MI355X_MICROARCH.md: one counter group per pass, --kernel-trace only) over a short bench run.
# usage (GPU box): bash tests/tools/pmc_step.sh edsr|rcan
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
M=${1:-edsr}
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmcs_${M}_$c -o p --output-format csv -- python3 $R/bench.py --model $M --steps 3 --warmup 1 --probe-steps 1 --no-cpu-baseline > /dev/null 2>&1
done
cd $R
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob('gpurun_out/pmcs_${M}_*')):
    for f in glob.glob(d+'/*counter_collection.csv'):
        by=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'rcab_kernel' in r['Kernel_Name'] or 'conv_block_kernel' in r['Kernel_Name']:
                by[(r['Kernel_Name'][:48], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k,v in sorted(by.items()):
            v=v[len(v)//2:]
            print(d, k, 'mean per launch %.1f KB' % (sum(v)/len(v)), 'n', len(v))
PY
