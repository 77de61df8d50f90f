#!/bin/bash
# SQ / LDS counters of the Cin = 256 data-gradient kernel (GPU box): bash tests/tools/pmc_conv4.sh   [env RUMPY_CONV4K=1 etc. select the kernel]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc4_*
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $R/gpurun_out/pmc4_a -o p --output-format csv -- python3 $R/tests/tools/kbench.py conv4 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --kernel-trace -d $R/gpurun_out/pmc4_b -o p --output-format csv -- python3 $R/tests/tools/kbench.py conv4 > /dev/null 2>&1
cd $R
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob('gpurun_out/pmc4_*')):
    for f in glob.glob(d+'/*counter_collection.csv'):
        rows=[r for r in csv.DictReader(open(f)) if 'conv4' in r['Kernel_Name'] or 'conv3x3_kernel<4' in r['Kernel_Name']]
        by=collections.defaultdict(list)
        for r in rows:
            if int(r['Grid_Size']) >= 256*200: by[(r['Kernel_Name'][:30], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k,v in sorted(by.items()):
            print(k[0], '%-28s mean per launch %14.0f  n %d' % (k[1], sum(v)/len(v), len(v)))
PY
