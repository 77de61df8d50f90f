#!/bin/bash
# same-box, same-library A/B of an environment switch: usage ab_env.sh "<bench args>" VAR=value [reps]
args=$1; kv=$2; reps=${3:-3}
for rep in $(seq $reps); do
  for mode in off on; do
    if [ $mode = on ]; then export $kv; else unset ${kv%%=*}; fi
    python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('%-4s %s: %9.1f patches/s  %.4f ms/step  kernel %.2f us' % ('$mode', '$kv', j['value'], j['ms_per_step'], j['roofline']['avg_launch_us']))"
  done
done
