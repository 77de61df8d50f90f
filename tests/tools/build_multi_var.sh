#!/bin/bash
# A variant build of the kernel library with SEVERAL sources recompiled with extra flags (a switch that lives in a shared header), the other objects from the
# product build.   usage: tests/tools/build_multi_var.sh <tag> "<flags>" <source.hip> [<source.hip> ...]   ->  build_abl/<tag>/librumpy_amd.so
set -e
root="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$root/rumpy_amd/csrc"
tag=$1; flags=$2; shift 2
make -s -j6
d="$root/build_abl/$tag"; rm -rf "$d"; mkdir -p "$d"
skip=""
for src in "$@"; do
  nopacked=""; case "$src" in conv_rcab*.hip) nopacked="-Xclang -target-feature -Xclang -packed-fp32-ops";; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $nopacked $flags -c "$src" -o "$d/${src%.hip}.o" 2>&1 | grep -v "not a recognized feature" || true
  skip="$skip|^${src%.hip}.o\$"
done
objs=$(ls *.o | grep -Ev "${skip#|}")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$d/librumpy_amd.so" $objs "$d"/*.o
rm "$d"/*.o
echo "$d/librumpy_amd.so"
