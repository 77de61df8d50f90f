#!/bin/bash
# A variant build of the kernel library from the CURRENT sources: one source file recompiled with extra flags, the other objects from the product build.
# usage: tests/tools/build_var.sh <source.hip> <tag> [flags...]   ->  build_abl/<tag>/librumpy_amd.so   (select with RUMPY_AMD_LIB=build_abl/<tag>/librumpy_amd.so)
# (stamps builds: -DC1_STAMPS / -DCHAIN_STAMPS; register / read-ahead experiments: -DC1_AHEAD=2 ...).  build_abl/ is git-ignored and travels to the GPU box.
set -e
root="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$root/rumpy_amd/csrc"
src=$1; tag=$2; shift 2
make -s -j6
d="$root/build_abl/$tag"; mkdir -p "$d"
nopacked=""; case "$src" in conv_rcab*.hip) nopacked="-Xclang -target-feature -Xclang -packed-fp32-ops";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $nopacked "$@" -c "$src" -o "$d/var.o"
objs=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$d/librumpy_amd.so" $objs "$d/var.o"
rm "$d/var.o"
echo "$d/librumpy_amd.so"
