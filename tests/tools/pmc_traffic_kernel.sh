#!/bin/bash
# HBM traffic of ONE kernel (name substring) inside the training step: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only) over a short bench run.
# usage (GPU box): [env switches of the engine] bash tests/tools/pmc_traffic_kernel.sh <substring> [model]       prints KB per launch; HBM read bytes = 2 x FETCH_SIZE on
# gfx950 for wide coalesced reads (MI355X_MICROARCH.md), LDS-DMA (global_load_lds_dwordx4) reads included
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; K=${1:-wgrad_dma}; M=${2:-edsr}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmct_$c
  rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmct_$c -o p --output-format csv -- python3 $R/bench.py --model $M --steps 3 --warmup 1 --probe-steps 1 --no-cpu-baseline --settle-ms 0 > /dev/null 2>&1
done
cd $R
python3 - <<PY
import csv, glob, collections
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob('gpurun_out/pmct_%s/*counter_collection.csv' % c):
        by = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if '$K' in r['Kernel_Name']:
                by[(r['Kernel_Name'][:48], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k, v in sorted(by.items()):
            v = v[len(v) // 2:]
            print(k[0], '%-12s mean per launch %12.1f KB  n %d' % (k[1], sum(v) / len(v), len(v)))
PY
