#!/bin/bash
# Round-4 baseline of a box: GPU suite, headline / RCAN bench lines, rocprofv3 kernel stats of the headline step.
# usage (GPU box): bash tests/tools/r04_base.sh <tag> [notest]   -> gpurun_out/base_<tag>/
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/base_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $R
if [ "$2" != notest ]; then timeout 900 python3 -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log; fi
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench_err.log
python3 bench.py --model rcan --no-cpu-baseline > $OUT/rcan_bench_line.json 2>> $OUT/bench_err.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_edsr -o p -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $OUT/prof_edsr.log 2>&1
cd $R
cp $(find $OUT/prof_edsr -name '*kernel_stats.csv' | head -1) $OUT/edsr_kernel_stats.csv; rm -rf $OUT/prof_edsr
python3 tests/tools/prof_summary.py $OUT/edsr_kernel_stats.csv 125 2>/dev/null | head -30
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/*bench_line.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        print(os.path.basename(f), 'NO LINE'); continue
    r = d.get('roofline') or {}
    print('%-30s %10.2f %-13s %8.3f ms  kernel %s us frac %s traffic %s' % (os.path.basename(f), d['value'], d['unit'], d['ms_per_step'], r.get('avg_launch_us'), r.get('frac'), r.get('traffic')))
PY
