#!/bin/bash
# one-rank RCCL step time of the data-parallel path, alternating: plain | collective from the current stream | side-stream form
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/dpab; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29598 $R/bench.py --gpus 1 --steps 300 --warmup 50 --probe-steps 0 --no-cpu-baseline "$@" 2>&1 | grep '^{"metric"' | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"]["loss"])'; }
for rep in 1 2 3; do
  echo "plain       $(python3 $R/bench.py --steps 300 --warmup 50 --probe-steps 0 --no-cpu-baseline "$@" | grep '^{"metric"' | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["config"]["loss"])')"
  echo "dp inline   $(RUMPY_DP_FORCE=1 RUMPY_DP_INLINE=1 run "$@")"
  echo "dp side     $(RUMPY_DP_FORCE=1 RUMPY_DP_INLINE=0 run "$@")"
done | tee $OUT/ab.txt
