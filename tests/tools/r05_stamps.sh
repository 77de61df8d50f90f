#!/bin/bash
# build (CPU box or GPU box) the -DRCAB2_STAMPS measurement library: bash tests/tools/r05_stamps.sh build ; on the GPU box: bash tests/tools/r05_stamps.sh run
root="$(cd "$(dirname "$0")/../.." && pwd)"
if [ "$1" = build ]; then
  set -e
  cd "$root/rumpy_amd/csrc" && make -s -j6
  d="$root/build_abl/R2_STAMPS"; mkdir -p $d
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Xclang -target-feature -Xclang -packed-fp32-ops -DRCAB2_STAMPS $EXTRA -c conv_rcab2.hip -o $d/abl.o 2>&1 | grep -v "not a recognized" || true
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/librumpy_amd.so $(ls *.o | grep -v "^conv_rcab2.o$") $d/abl.o
  rm $d/abl.o
else
  cd "$root" && RUMPY_AMD_LIB=$root/build_abl/R2_STAMPS/librumpy_amd.so python3 tests/tools/kbench.py rcab2stamps
fi
