#!/bin/bash
# round 3: the column-tiled one-launch kernels - parity tests, then the bench lines of the reference's own shapes with their A/B
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r03_wide}; mkdir -p $OUT
cd $R
python -m pytest tests/test_kernels_gpu.py -x -q -k "conv_block" > $OUT/t_kernels.log 2>&1; echo "kernels rc=$?" | tee -a $OUT/summary
python -m pytest tests/test_network_gpu.py -x -q -k "wide_image or shipped_crop or one_launch_rcab_matches or arbitrary or kept_on_device or starts_its_own or two_launch_path" > $OUT/t_network.log 2>&1; echo "network rc=$?" | tee -a $OUT/summary
python -m pytest tests/test_contrastive_gpu.py -x -q -k "survives or bitwise_reproducible" > $OUT/t_contrastive.log 2>&1; echo "contrastive rc=$?" | tee -a $OUT/summary
tail -3 $OUT/t_kernels.log $OUT/t_network.log $OUT/t_contrastive.log
b() { python bench.py --no-cpu-baseline "$@" 2>&1 | grep '^{"metric"'; }
for rep in 1 2; do
  b --steps 200 --warmup 30 > $OUT/edsr48_$rep.json
  b --lr-size 64 --batch 16 --steps 100 --warmup 20 > $OUT/edsr64_$rep.json
  RUMPY_BLOCK_W48=1 b --lr-size 64 --batch 16 --steps 100 --warmup 20 > $OUT/edsr64_w48_$rep.json
  b --model rcan --lr-size 64 --batch 8 --steps 30 --warmup 5 > $OUT/rcan64_$rep.json
  RUMPY_BLOCK_W48=1 b --model rcan --lr-size 64 --batch 8 --steps 30 --warmup 5 > $OUT/rcan64_w48_$rep.json
done
python bench.py --mode eval --steps 50 --warmup 5 2>&1 | grep '^{"metric"' > $OUT/eval_edsr.json
python bench.py --mode eval --model rcan --steps 20 --warmup 3 2>&1 | grep '^{"metric"' > $OUT/eval_rcan.json
python - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), 'NO LINE'); continue
    r = d.get('roofline') or {}
    print('%-22s %10.2f %-12s %8.3f ms  kernel %s us frac %s  %s' % (os.path.basename(f), d['value'], d['unit'], d['ms_per_step'], r.get('avg_launch_us'), r.get('frac'),
          {k: v for k, v in d['config'].items() if k.startswith('ms_')}))
PY
