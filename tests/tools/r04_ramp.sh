#!/bin/bash
# how the driver's short run (--steps 20 --warmup 5) relates to the long ones: warm-up dependence of the step time
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/ramp_${1:-a}; rm -rf $OUT; mkdir -p $OUT; cd $R
for rep in 1 2; do
  for cfg in "20 5" "20 100" "200 5" "200 30" "20 5"; do
    set -- $cfg
    python3 bench.py --no-cpu-baseline --steps $1 --warmup $2 > $OUT/s$1_w$2_$rep.json 2>> $OUT/err.log
    python3 - <<PY
import json
d = json.loads(open('$OUT/s$1_w$2_$rep.json').read().strip().splitlines()[-1])
print('steps %4d warmup %4d : %9.1f patches/s %7.4f ms  block launch %s us' % ($1, $2, d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_us')))
PY
  done
done
python3 bench.py --steps 20 --warmup 5 > $OUT/default_cpu.json 2>> $OUT/err.log
python3 - <<PY
import json
d = json.loads(open('$OUT/default_cpu.json').read().strip().splitlines()[-1])
print('with cpu baseline leg: %9.1f patches/s %7.4f ms' % (d['value'], d['ms_per_step']))
PY
