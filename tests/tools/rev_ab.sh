#!/bin/bash
# Same-box A/B of two TREES (not just two libraries): the current one against a checked-out revision with its own host code, built in build_abl/<dir>
#   git archive <rev> | tar -x -C build_abl/<dir>; (cd build_abl/<dir>; python -c "import __graft_entry__ as g; g.build()")
# usage (GPU box): bash tests/tools/rev_ab.sh <dir> [reps] ["<bench.py args>"]
cd ${GRAFT_REPO_ROOT:-.}; R=$(pwd); D=$R/build_abl/${1:-r05_tree}; reps=${2:-3}; args=${3:---no-as-called --no-cpu-baseline}
O=$R/gpurun_out/rev_ab; rm -rf $O; mkdir -p $O
line() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline') or {}
print('%-8s %10.1f %s  %.4f ms/step  settled %s  dominant kernel %s us' % (sys.argv[2], d['value'], d['unit'], d['ms_per_step'], (d.get('settled') or {}).get('value'), r.get('avg_launch_us')))" $1 $2; }
for i in $(seq $reps); do
  ( cd $D && python3 bench.py $args > $O/old_$i.json 2>> $O/err.log ); line $O/old_$i.json old
  ( cd $R && python3 bench.py $args > $O/new_$i.json 2>> $O/err.log ); line $O/new_$i.json new
done
tail -3 $O/err.log
