# SQ / LDS counters of the block kernels (conv_block_kernel / rcab_kernel) INSIDE the training step: one rocprofv3 --pmc pass over a short bench run
# usage (GPU box): [SQ_KERNELS='wgrad_dma_kernel conv4dt_kernel ...'] bash tests/tools/pmc_step_sq.sh edsr|rcan [fp8]      (SQ_KERNELS: other kernels of the step, by substring)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
M=${1:-edsr}
PREC=${2:-bf16}
rm -rf $R/gpurun_out/pmcq_${M}
# (the fp8 kernels are conv_block_fp8_kernel / rcab_fp8_kernel: matched by the same substrings below)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $R/gpurun_out/pmcq_${M} -o p --output-format csv -- python3 $R/bench.py --model $M --precision $PREC --steps 3 --warmup 1 --probe-steps 1 --no-cpu-baseline --settle-ms 0 > /dev/null 2>&1
cd $R
python3 - <<PY
import csv,glob,collections,os
NAMES = tuple(os.environ.get('SQ_KERNELS', '').split()) or ('rcab_kernel', 'rcab2_kernel', 'conv_block_kernel', 'block_chain_kernel', 'rcab_fp8_kernel', 'conv_block_fp8_kernel')
for f in glob.glob('gpurun_out/pmcq_${M}/*counter_collection.csv'):
    by=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if any(t in r['Kernel_Name'] for t in NAMES):
            by[(r['Kernel_Name'][:48], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(by.items()):
        v=v[len(v)//2:]
        print(k[0], '%-26s mean per launch %14.0f  n %d' % (k[1], sum(v)/len(v), len(v)))
for f in glob.glob('gpurun_out/pmcq_${M}/*kernel_trace.csv'):
    by=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if any(t in r['Kernel_Name'] for t in NAMES):
            by[r['Kernel_Name'][:48]].append(float(r['End_Timestamp'])-float(r['Start_Timestamp']))
    for k,v in sorted(by.items()):
        v=v[len(v)//2:]
        print(k, 'duration under the counter pass: mean %.2f us' % (sum(v)/len(v)/1e3))
PY
