#!/bin/bash
# Timing experiments: builds of the kernel library with one piece of a kernel compiled out (results are WRONG).
# usage: [EXTRA='-DOTHER=1' TAG=_x] tests/tools/build_abl.sh <source.hip> <MACRO> <values...>   ->  build_abl/<MACRO>_<v>[TAG]/librumpy_amd.so
# Round 4: the *_ABL branches no longer live in the shipped sources.  The experiment source is compiled from a scratch copy of rumpy_amd/csrc
# with tests/tools/patches/abl_r03.patch applied (the patch matches the sources of the commit that introduced it; later kernel edits may need
# a rebase of the hunk they touch - `git apply --3way`).
set -e
root="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$root/rumpy_amd/csrc"
src=$1; macro=$2; shift 2
make -s -j6
scratch="$root/build_abl/_src"; rm -rf "$scratch"; mkdir -p "$scratch/rumpy_amd" "$scratch/include"
cp -r "$root/rumpy_amd/csrc" "$scratch/rumpy_amd/csrc"; cp "$root"/include/*.h "$scratch/include/"; rm -f "$scratch"/rumpy_amd/csrc/*.o
(cd "$scratch" && git init -q . 2>/dev/null; git -C "$scratch" apply "$root/tests/tools/patches/abl_r03.patch")
for v in "$@"; do
  d=../../build_abl/${macro}_$v$TAG; mkdir -p $d
  src_dir="$scratch/rumpy_amd/csrc"
  (cd "$src_dir" && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $( [ "$src" = conv_rcab.hip ] && echo -Xclang -target-feature -Xclang -packed-fp32-ops ) -D${macro}=$v $EXTRA -c $src -o "$root/build_abl/${macro}_$v$TAG/abl.o")
  objs=$(ls *.o | grep -v "^${src%.hip}.o$")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/librumpy_amd.so $objs $d/abl.o
  rm $d/abl.o
done
