#!/bin/bash
# Round 5, first call: BASELINE config 5 as named (full-depth blind QRCAN, precision='fp8') against the fp32 oracle, the small-gradient scale
# test, and this box's RCAN baseline lines for later A/B.     usage (GPU box): bash tests/tools/r05_pin.sh   -> gpurun_out/r05_pin/
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_pin
rm -rf $OUT; mkdir -p $OUT
cd $R
python3 -m pytest tests/test_fp8_gpu.py -x -q -s -k "config5 or mean_reduced" > $OUT/pin_tests.log 2>&1
echo "pin tests rc $?" >> $OUT/pin_tests.log
python3 -m pytest tests/test_fp8_gpu.py -x -q > $OUT/fp8_all.log 2>&1
echo "fp8 all rc $?" >> $OUT/fp8_all.log
python3 bench.py --model rcan --no-cpu-baseline > $OUT/rcan_line.json 2> $OUT/err.log
python3 bench.py --no-cpu-baseline > $OUT/edsr_line.json 2>> $OUT/err.log
python3 tests/tools/kbench.py rcab > $OUT/kbench_rcab.txt 2>&1
tail -25 $OUT/pin_tests.log; tail -5 $OUT/fp8_all.log; cat $OUT/kbench_rcab.txt | tail -5
python3 - <<PY
import json
for f in ('rcan_line.json','edsr_line.json'):
    d = json.loads(open('$OUT/'+f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d.get('cold_start'))
PY
