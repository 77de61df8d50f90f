#!/bin/bash
# round 3: weight-gradient kernel, DMA issue staggered between the two waves of a SIMD - same-box A/B (GPU box): bash tests/tools/r03_wgrad.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_wgrad; mkdir -p $O; cd $R
python -m pytest tests/test_kernels_gpu.py -q -x -k "wgrad" > $O/pytest_wgrad.txt 2>&1; tail -2 $O/pytest_wgrad.txt
python -m pytest tests/test_network_gpu.py -q -x -k "gradient_parity or bitwise or determin" > $O/pytest_net.txt 2>&1; tail -2 $O/pytest_net.txt
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2 3; do
  echo -n "staggered: "; python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | line
  echo -n "all waves behind the barrier: "; RUMPY_AMD_LIB=$R/build_abl/WGRAD_STAGGER_0/librumpy_amd.so python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | line
done > $O/bench_ab.txt 2>&1
cat $O/bench_ab.txt
for v in "" "RUMPY_AMD_LIB=$R/build_abl/WGRAD_STAGGER_0/librumpy_amd.so"; do
  echo "== kernel stats $v"
  ( export $v; bash tests/tools/kstat.sh "--steps 40 --warmup 5" "wgrad|conv_block" 6 )
done > $O/kstat_ab.txt 2>&1
cat $O/kstat_ab.txt
echo "== rcan"; for i in 1 2; do
  echo -n "staggered: "; python bench.py --model rcan --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | line
  echo -n "barrier: "; RUMPY_AMD_LIB=$R/build_abl/WGRAD_STAGGER_0/librumpy_amd.so python bench.py --model rcan --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | line
done > $O/bench_rcan_ab.txt 2>&1
cat $O/bench_rcan_ab.txt
