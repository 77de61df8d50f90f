#!/bin/bash
# Round 5: the exchange-free RCAB launches (conv_rcab2.hip) - kernel parity, network tests on both forms, same-box A/B against conv_rcab.hip.
# usage (GPU box): bash tests/tools/r05_rcab2.sh [quick]   -> gpurun_out/r05_rcab2/
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_rcab2
rm -rf $OUT; mkdir -p $OUT
cd $R
python3 -m pytest tests/test_rcab2_gpu.py -x -q > $OUT/kernel_tests.log 2>&1; echo "rc $?" >> $OUT/kernel_tests.log
tail -15 $OUT/kernel_tests.log
python3 tests/tools/kbench.py rcab > $OUT/kbench.txt 2>&1
python3 tests/tools/kbench.py rcab2 >> $OUT/kbench.txt 2>&1
grep "us per launch" $OUT/kbench.txt
python3 -m pytest tests/test_network_gpu.py tests/test_qrcan_gpu.py -x -q -k "rcab or rcan or qrcan" > $OUT/net_tests.log 2>&1; echo "rc $?" >> $OUT/net_tests.log
tail -15 $OUT/net_tests.log
for rep in 1 2; do
  for form in lazy xchg; do
    RUMPY_RCAB_FORM=$form python3 bench.py --model rcan --no-cpu-baseline --steps 100 --warmup 20 > $OUT/rcan_${form}_$rep.json 2>> $OUT/err.log
  done
done
python3 - <<PY
import json
for rep in (1, 2):
    for form in ('lazy', 'xchg'):
        try:
            d = json.loads(open('$OUT/rcan_%s_%d.json' % (form, rep)).read().strip().splitlines()[-1])
            print(form, rep, d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('avg_launch_us'))
        except Exception as e:
            print(form, rep, 'NO LINE', e)
PY
tail -5 $OUT/err.log
