#!/bin/bash
# RCAB strip heights: tests, then the reference's RCAN crop shape per forced geometry.   usage: bash tests/tools/r04_geo2.sh <tag>
TAG=${1:-a}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/geo2_$TAG; rm -rf $OUT; mkdir -p $OUT; cd $R
timeout 900 python3 -m pytest tests/test_network_gpu.py tests/test_qrcan_gpu.py -x -q -k "rcab or rcan or qrcan" > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for geo in auto 6,2 4,2 8,2; do
  if [ $geo = auto ]; then unset RUMPY_BLOCK_GEO; else export RUMPY_BLOCK_GEO=$geo; fi
  python3 bench.py --model rcan --lr-size 64 --batch 8 --steps 40 --warmup 8 --no-cpu-baseline > $OUT/rcan64_$geo.json 2>> $OUT/err.log
done
unset RUMPY_BLOCK_GEO
python3 bench.py --model rcan --no-cpu-baseline --steps 100 --warmup 20 > $OUT/rcan48_bf16.json 2>> $OUT/err.log
python3 bench.py --model rcan --no-cpu-baseline --steps 100 --warmup 20 --precision fp8 > $OUT/rcan48_fp8.json 2>> $OUT/err.log
python3 bench.py --model blindqrcan --no-cpu-baseline --steps 100 --warmup 20 > $OUT/blind48_bf16.json 2>> $OUT/err.log
python3 bench.py --model blindqrcan --no-cpu-baseline --steps 100 --warmup 20 --precision fp8 > $OUT/blind48_fp8.json 2>> $OUT/err.log
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
tail -3 $OUT/err.log
