#!/bin/bash
# per-dispatch kernel durations of a few EDSR training steps (rocprofv3 --kernel-trace), summarised per (kernel, grid) by tests/tools/step_trace.py
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --model ${1:-edsr} --steps 20 --warmup 5 --probe-steps 0 --no-cpu-baseline > $OUT/bench.log 2>&1
python3 tests/tools/step_trace.py $OUT
