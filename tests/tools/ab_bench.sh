#!/bin/bash
# Same-box A/B of whole training steps: bench.py with each of the given libraries (paths under build_abl/, or "tree" = the in-tree build),
# alternating, REPS times.  usage: tests/tools/ab_bench.sh "<bench args>" <lib> <lib> ...
args=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = tree ]; then unset RUMPY_AMD_LIB; else export RUMPY_AMD_LIB=$PWD/build_abl/$lib/librumpy_amd.so; fi
    python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('%-12s %s: %9.1f patches/s  %.4f ms/step  kernel %.2f us  frac %.4f' % ('$lib', '$args', j['value'], j['ms_per_step'], j['roofline']['avg_launch_us'], j['roofline']['frac']))"
  done
done
