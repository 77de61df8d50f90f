#!/bin/bash
# round 3: weight-gradient kernel with fixed roles (multiplier / loader waves) against the all-purpose waves (RUMPY_WGRAD_NO_ROLES=1): bash tests/tools/r03_wgrad2.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_wgrad2; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "wgrad" > $O/pytest_wgrad.txt 2>&1; tail -n 3 $O/pytest_wgrad.txt
timeout 900 python -m pytest tests/test_network_gpu.py -q -x -k "gradient_parity or bitwise or determin or small_train" > $O/pytest_net.txt 2>&1; tail -n 3 $O/pytest_net.txt
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  echo -n "roles: "; python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | line
  echo -n "all-purpose waves: "; RUMPY_WGRAD_NO_ROLES=1 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | line
done > $O/bench_ab.txt 2>&1
cat $O/bench_ab.txt
for v in "X=1" "RUMPY_WGRAD_NO_ROLES=1"; do
  echo "== kernel stats $v"
  ( export $v; bash tests/tools/kstat.sh "--steps 40 --warmup 5" "wgrad" 4 )
done 2>&1 | grep -v "^declare" > $O/kstat_ab.txt
cat $O/kstat_ab.txt
echo "== rcan"; for i in 1 2; do
  echo -n "roles: "; python bench.py --model rcan --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | line
  echo -n "all-purpose: "; RUMPY_WGRAD_NO_ROLES=1 python bench.py --model rcan --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | line
done > $O/bench_rcan_ab.txt 2>&1
cat $O/bench_rcan_ab.txt
