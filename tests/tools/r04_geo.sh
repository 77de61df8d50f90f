#!/bin/bash
# strip heights: tests, then 64-pixel training and whole-image evaluation lines per forced geometry.   usage: bash tests/tools/r04_geo.sh <tag>
TAG=${1:-a}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/geo_$TAG; rm -rf $OUT; mkdir -p $OUT; cd $R
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "conv_block" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for geo in auto 6,3 6,2 8,2 4,2; do
  if [ $geo = auto ]; then unset RUMPY_BLOCK_GEO; else export RUMPY_BLOCK_GEO=$geo; fi
  python3 bench.py --lr-size 64 --batch 16 --steps 200 --warmup 30 --no-cpu-baseline > $OUT/edsr64_$geo.json 2>> $OUT/err.log
  python3 bench.py --mode eval --steps 60 --warmup 6 --no-cpu-baseline > $OUT/eval_$geo.json 2>> $OUT/err.log
done
unset RUMPY_BLOCK_GEO
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        print(os.path.basename(f), 'NO LINE'); continue
    r = d.get('roofline') or {}
    print('%-24s %10.2f %-13s %8.3f ms  kernel %s us frac %s' % (os.path.basename(f), d['value'], d['unit'], d['ms_per_step'], r.get('avg_launch_us'), r.get('frac')))
PY
