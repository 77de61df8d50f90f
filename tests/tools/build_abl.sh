#!/bin/bash
# Timing experiments: builds of the kernel library with one piece of a kernel compiled out (results are WRONG).
# usage: [EXTRA='-DOTHER=1' TAG=_x] tests/tools/build_abl.sh <source.hip> <MACRO> <values...>   ->  build_abl/<MACRO>_<v>[TAG]/librumpy_amd.so
set -e
cd "$(dirname "$0")/../../rumpy_amd/csrc"
src=$1; macro=$2; shift 2
make -s -j6
for v in "$@"; do
  d=../../build_abl/${macro}_$v$TAG; mkdir -p $d
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $( [ "$src" = conv_rcab.hip ] && echo -Xclang -target-feature -Xclang -packed-fp32-ops ) -D${macro}=$v $EXTRA -c $src -o $d/abl.o
  objs=$(ls *.o | grep -v "^${src%.hip}.o$")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/librumpy_amd.so $objs $d/abl.o
  rm $d/abl.o
done
