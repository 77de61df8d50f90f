#!/bin/bash
# SQ counters of one kernel (name substring) inside the training step: two rocprofv3 --pmc passes over a short bench run
# usage (GPU box): bash tests/tools/pmc_kernel.sh <substring> [model]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; K=${1:-wgrad_dma}; M=${2:-edsr}
rm -rf $R/gpurun_out/pmck_a $R/gpurun_out/pmck_b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $R/gpurun_out/pmck_a -o p --output-format csv -- python3 $R/bench.py --model $M --steps 3 --warmup 1 --probe-steps 1 --no-cpu-baseline --settle-ms 0 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU --kernel-trace -d $R/gpurun_out/pmck_b -o p --output-format csv -- python3 $R/bench.py --model $M --steps 3 --warmup 1 --probe-steps 1 --no-cpu-baseline --settle-ms 0 > /dev/null 2>&1
cd $R
python3 - <<PY
import csv,glob,collections
for d in ('pmck_a','pmck_b'):
    for f in glob.glob('gpurun_out/%s/*counter_collection.csv' % d):
        by=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if '$K' in r['Kernel_Name']:
                by[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k,v in sorted(by.items()):
            v=v[len(v)//2:]
            print(k[0], '%-26s mean per launch %14.0f  n %d' % (k[1], sum(v)/len(v), len(v)))
PY
