#!/bin/bash
# losses of short bench runs: plain (with the plan form of data-parallel runs, RUMPY_WGRAD_AB=1: same jobs) twice, then one rank over RCCL (RUMPY_DP_FORCE=1) with and without the one-launch RCAB kernels
T="--model rcan --steps 6 --warmup 2 --probe-steps 2 --no-cpu-baseline"
g() { grep -o '"loss": [0-9.e-]*'; }
echo plain1; RUMPY_WGRAD_AB=1 python bench.py $T 2>&1 | g
echo plain2; RUMPY_WGRAD_AB=1 python bench.py $T 2>&1 | g
echo plain_noprobe; RUMPY_WGRAD_AB=1 python bench.py --model rcan --steps 6 --warmup 2 --probe-steps 0 --no-cpu-baseline 2>&1 | g
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo dp; RUMPY_DP_FORCE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29581 bench.py --gpus 1 $T 2>&1 | g
echo dp2; RUMPY_DP_FORCE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29582 bench.py --gpus 1 $T 2>&1 | g
echo dp_norcab; RUMPY_NO_RCAB=1 RUMPY_DP_FORCE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29583 bench.py --gpus 1 $T 2>&1 | g
echo plain_norcab; RUMPY_WGRAD_AB=1 RUMPY_NO_RCAB=1 python bench.py $T 2>&1 | g
