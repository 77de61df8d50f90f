#!/bin/bash
# same-box A/B of the ablation builds of conv_mfma.hip's Cin = 256 kernel: usage abl_conv4.sh <dir names under build_abl/>
echo "== base"; python tests/tools/kbench.py conv4 2>&1 | grep -v amdgpu.ids
for v in "$@"; do echo "== $v"; RUMPY_AMD_LIB=$PWD/build_abl/$v/librumpy_amd.so python tests/tools/kbench.py conv4 2>&1 | grep -v amdgpu.ids; done
