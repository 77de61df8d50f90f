#!/bin/bash
# same-box A/B of the ablation / policy builds of conv_block.hip (tests/tools/build_abl.sh): usage abl_block.sh <dir names under build_abl/>
export KBENCH_FUSED_ONLY=1
echo "== base"; python tests/tools/kbench.py block 2>&1 | grep "residual"
for v in "$@"; do echo "== $v"; RUMPY_AMD_LIB=$PWD/build_abl/$v/librumpy_amd.so python tests/tools/kbench.py block 2>&1 | grep -v amdgpu.ids; done
echo "== base again"; python tests/tools/kbench.py block 2>&1 | grep "residual"
