#!/bin/bash
# round 3: block-kernel experiments - parity tests, then same-box A/B of ablation builds (kbench block) and of the whole step (bench.py)
# usage: bash tests/tools/r03_block.sh <out dir name> [build_abl/<dir> ...]
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r03_block}; mkdir -p $OUT; cd $R
shift
python -m pytest tests/test_kernels_gpu.py -x -q -k "conv_block" > $OUT/t_kernels.log 2>&1; echo "kernels rc=$?" | tee $OUT/summary
python -m pytest tests/test_network_gpu.py -x -q -k "edsr_small or edsr_baseline_full_depth or two_launch_path or determinism" > $OUT/t_network.log 2>&1; echo "network rc=$?" | tee -a $OUT/summary
tail -n 2 $OUT/t_kernels.log $OUT/t_network.log
export KBENCH_FUSED_ONLY=1
for rep in 1 2; do
  echo "== base"; python tests/tools/kbench.py block 2>&1 | grep "residual" | tail -1
  for v in "$@"; do echo "== $v"; RUMPY_AMD_LIB=$PWD/build_abl/$v/librumpy_amd.so python tests/tools/kbench.py block 2>&1 | grep -v amdgpu.ids | tail -${KB_TAIL:-1}; done
done | tee $OUT/kbench.txt
b() { python bench.py --no-cpu-baseline --steps 300 --warmup 40 2>&1 | grep '^{"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"; }
for rep in 1 2; do
  echo "base: $(b)"
  for v in "$@"; do case $v in *ABL_9*) continue;; esac; echo "$v: $(RUMPY_AMD_LIB=$PWD/build_abl/$v/librumpy_amd.so b)"; done
done | tee $OUT/bench.txt
