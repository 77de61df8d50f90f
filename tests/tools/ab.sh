#!/bin/bash
# ONE same-box A/B runner (round 6: replaces the one-letter scripts r05_[a-z].sh / r06_[a-i].sh - their arms are the table in tests/tools/README.md).
#
#   bash tests/tools/ab.sh [-n NAME] [-r REPS] [-b "<bench.py args>"] [-t "<pytest args>"] [-k] ARM [ARM ...]
#
# ARM = label[:VAR=value[,VAR=value...]][@TAG]   one arm of the comparison: environment switches of the engine (RUMPY_NO_CHAIN=1, RUMPY_RCAB_FORM=lazy ...)
#                                                and / or a variant build of the library (build_abl/TAG/librumpy_amd.so, made by build_var.sh / build_abl.sh)
#   -r REPS   the arms run alternating REPS times (default 3): one bench.py line per arm and repetition -> value, settled value, dominant-kernel time
#   -b ARGS   bench.py arguments (default: --no-as-called --no-cpu-baseline, i.e. the headline workload, 400 steps)
#   -t ARGS   first run `pytest ARGS` once per arm (parity of a variant build before it is timed)
#   -k        after the timing, one rocprofv3 --kernel-trace --stats pass per arm (60 steps): the seven largest kernels of the step
#   -n NAME   output directory gpurun_out/ab_NAME (default: ab)
# Example (round 6, the weight-gradient pin):  bash tests/tools/build_var.sh wgrad_dma.hip WGRAD_PIN0 -DWGRAD_PIN=0
#                                              gpurun -- 'bash tests/tools/ab.sh -n wgrad_pin -k -t "tests/test_kernels_gpu.py -q -k wgrad" pin in_front@WGRAD_PIN0'
cd ${GRAFT_REPO_ROOT:-.}; R=$(pwd)
name=ab; reps=3; bargs="--no-as-called --no-cpu-baseline"; targs=""; kstats=0
while getopts "n:r:b:t:k" o; do case $o in n) name=$OPTARG;; r) reps=$OPTARG;; b) bargs=$OPTARG;; t) targs=$OPTARG;; k) kstats=1;; esac; done
shift $((OPTIND - 1))
O=$R/gpurun_out/ab_$name; rm -rf $O; mkdir -p $O
arm_env() {      # prints the `env` assignments of an arm
  local arm=$1 spec lib=""
  spec=${arm#*:}; [ "$spec" = "$arm" ] && spec=""
  case $arm in *@*) lib=${arm##*@}; spec=${spec%@*};; esac
  [ -n "$lib" ] && echo "RUMPY_AMD_LIB=$R/build_abl/$lib/librumpy_amd.so"
  [ -n "$spec" ] && echo "$spec" | tr ',' '\n'
}
label() { local a=${1%%:*}; echo ${a%%@*}; }
if [ -n "$targs" ]; then
  for arm in "$@"; do
    ( export $(arm_env $arm) >/dev/null 2>&1; timeout 1500 python3 -m pytest $targs > $O/tests_$(label $arm).log 2>&1; echo "tests $(label $arm): $(tail -1 $O/tests_$(label $arm).log)" )
  done
fi
for rep in $(seq $reps); do
  for arm in "$@"; do
    ( export $(arm_env $arm) >/dev/null 2>&1; python3 bench.py $bargs > $O/$(label $arm)_$rep.json 2>> $O/err.log
      python3 - <<PY
import json
try:
    d = json.loads(open('$O/$(label $arm)_$rep.json').read().strip().splitlines()[-1]); r = d.get('roofline') or {}
    print('%-24s %10.1f %s   settled %s   %s us per launch of the dominant kernel' % ('$(label $arm)', d['value'], d['unit'], (d.get('settled') or {}).get('value'), r.get('avg_launch_us')))
except Exception as e:
    print('$(label $arm)', 'NO LINE', e)
PY
    )
  done
done
if [ $kstats = 1 ]; then
  cd /tmp && export TMPDIR=/tmp
  for arm in "$@"; do
    ( export $(arm_env $arm) >/dev/null 2>&1; rm -rf $O/prof
      rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 $R/bench.py ${bargs/--steps [0-9]*/} --steps 60 --warmup 20 --no-cpu-baseline --no-as-called --settled-probe-ms 0 > $O/prof_$(label $arm).log 2>&1
      echo "== $(label $arm)"; python3 $R/tests/tools/prof_summary.py $(find $O/prof -name '*kernel_stats.csv' | head -1) 1 | sort -k1,1 -n -r | head -8 )
  done
fi
tail -3 $O/err.log 2>/dev/null
