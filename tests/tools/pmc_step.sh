# HBM traffic of the block kernels (conv_block_kernel / rcab_kernel) from PMC counters, measured INSIDE the training step: separate
# rocprofv3 --pmc passes (MI355X_MICROARCH.md: one counter group per pass, --kernel-trace only) over a short bench run.
# usage (GPU box): bash tests/tools/pmc_step.sh edsr|rcan [fp8]      (fp8: the conv_block_fp8_kernel / rcab_fp8_kernel entries)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
M=${1:-edsr}
PREC=${2:-bf16}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmcs_${M}_$c
  rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmcs_${M}_$c -o p --output-format csv -- python3 $R/bench.py --model $M --precision $PREC --steps 3 --warmup 1 --probe-steps 1 --no-cpu-baseline --settle-ms 0 > /dev/null 2>&1
done
cd $R
python3 - <<PY
import csv, glob, collections, json, os, sys
sys.path.insert(0, '$R')
import bench
M = '${M}'
FP8 = '${PREC}' == 'fp8'
means = collections.defaultdict(dict)
for d in sorted(glob.glob('gpurun_out/pmcs_%s_*' % M)):
    for f in glob.glob(d + '/*counter_collection.csv'):
        by = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if any(t in r['Kernel_Name'] for t in (('rcab_fp8_kernel', 'conv_block_fp8_kernel') if FP8 else ('rcab_kernel', 'rcab2_kernel', 'conv_block_kernel', 'block_chain_kernel'))):
                by[(r['Kernel_Name'][:64], r['Counter_Name'])].append(float(r['Counter_Value']))
        for k, v in sorted(by.items()):
            v = v[len(v) // 2:]
            print(d, k, 'mean per launch %.1f KB' % (sum(v) / len(v)), 'n', len(v))
            means[k[0]][k[1]] = sum(v) / len(v)
# per kernel form: 2 x FETCH_SIZE (gfx950 reports half of wide coalesced reads) + WRITE_SIZE, KB of 1024 B; entry = mean over the forms (a step
# launches the forward and the data-gradient form equally often)
forms = {k: (2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024 for k, c in means.items() if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c}
if forms:
    lazy = any('rcab2_kernel' in k for k in forms)        # the default RCAB form since round 5 (conv_rcab2.hip); RUMPY_RCAB_FORM=xchg: conv_rcab.hip
    chain = any('block_chain_kernel' in k for k in forms)  # EDSR since round 5: the run of residual blocks is one persistent launch (conv_chain.hip); RUMPY_NO_CHAIN=1: conv_block.hip
    kind = ('rcab_fp8_kernel' if FP8 else 'rcab2_kernel' if lazy else 'rcab_kernel') if M != 'edsr' else ('conv_block_fp8_kernel' if FP8 else 'block_chain_kernel' if chain else 'conv_block_kernel')
    srcs = ['rumpy_amd/csrc/block_common.hpp'] + (['rumpy_amd/csrc/fp8_common.hpp'] if FP8 else [])
    if M != 'edsr':
        srcs += ['rumpy_amd/csrc/rcab_common.hpp', 'rumpy_amd/csrc/conv_rcab_fp8.hip' if FP8 else 'rumpy_amd/csrc/conv_rcab2.hip' if lazy else 'rumpy_amd/csrc/conv_rcab.hip']
    else:
        srcs += ['rumpy_amd/csrc/conv_block_fp8.hip'] if FP8 else ['rumpy_amd/csrc/chain_common.hpp', 'rumpy_amd/csrc/conv_chain.hip'] if chain else ['rumpy_amd/csrc/conv_block.hip']
    entry = {'bytes_per_launch': sum(forms.values()) / len(forms), 'per_form_bytes': forms, 'sources': srcs, 'sha16': bench.source_sha16(srcs),
             'source': 'profiles/pmc_traffic.json <- tests/tools/pmc_step.sh %s: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py --model %s '
                       '(32 x 48 x 48)%s, 2 x FETCH_SIZE + WRITE_SIZE, mean over the forward and data-gradient launches (block_chain_kernel: per launch of 16 blocks)' % (M, M, ' --precision fp8' if FP8 else '')}
    out = 'gpurun_out/pmc_traffic_%s%s.json' % (M, '_fp8' if FP8 else '')
    json.dump({'%s:N32:P48' % kind: entry}, open(out, 'w'), indent=1)
    print('wrote', out, '(merge into profiles/pmc_traffic.json)')
PY
