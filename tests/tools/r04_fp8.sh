#!/bin/bash
# fp8 opt-in: tests, then bf16 / fp8 bench lines of one model on one box, alternating.   usage: bash tests/tools/r04_fp8.sh <tag> <model> [notest]
TAG=${1:-a}; MODEL=${2:-edsr}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/fp8_$TAG; rm -rf $OUT; mkdir -p $OUT; cd $R
if [ "$3" != notest ]; then timeout 900 python3 -m pytest tests/test_fp8_gpu.py -x -q -s > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log; tail -15 $OUT/pytest.log; fi
for rep in 1 2; do
  python3 bench.py --model $MODEL --no-cpu-baseline --steps 200 --warmup 30 > $OUT/${MODEL}_bf16_$rep.json 2>> $OUT/err.log
  python3 bench.py --model $MODEL --no-cpu-baseline --steps 200 --warmup 30 --precision fp8 > $OUT/${MODEL}_fp8_$rep.json 2>> $OUT/err.log
done
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        print(os.path.basename(f), 'NO LINE'); continue
    r = d.get('roofline') or {}
    print('%-24s %10.2f %-13s %8.3f ms  kernel %s us frac %s loss %s' % (os.path.basename(f), d['value'], d['unit'], d['ms_per_step'], r.get('avg_launch_us'), r.get('frac'), d['config'].get('loss')))
PY
tail -5 $OUT/err.log
