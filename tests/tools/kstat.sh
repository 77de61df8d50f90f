#!/bin/bash
# per-kernel time of a short bench run (GPU box): bash tests/tools/kstat.sh "<bench args>" [name filter]
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/kstat; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $R/bench.py $1 --no-cpu-baseline > $OUT/log 2>&1
cd $R
python tests/tools/prof_summary.py $(find $OUT -name '*kernel_stats.csv' | head -1) 1 | sort -k1,1 -n -r | awk '{ $1=""; print }' | grep -E "${2:-.}" | head -${3:-30}
