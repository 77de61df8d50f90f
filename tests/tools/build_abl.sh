#!/bin/bash
# Timing experiments: builds of the kernel library with one piece of a kernel compiled out (results are WRONG).
# usage: [EXTRA='-DOTHER=1' TAG=_x] tests/tools/build_abl.sh <source.hip> <MACRO> <values...>   ->  build_abl/<MACRO>_<v>[TAG]/librumpy_amd.so
# Round 4: the *_ABL branches no longer live in the shipped sources.  The experiment source is compiled from a scratch copy of the kernel sources
# AS OF THE COMMIT THE PATCH WAS CUT FROM (ABL_BASE below; `git archive`, so this runs in the development container, where the history is - the
# libraries it builds travel to the GPU box) with tests/tools/patches/abl_r03.patch applied: the patch records the experiments of rounds 1-3 on
# the kernels of that time, it is not rebased onto later kernel edits (round 5, ADVICE r4: it had stopped applying to HEAD, and `--3way` cannot
# work in a fresh scratch tree).  The OTHER objects of the library come from the current build (the C ABI only grew since).
# tests/test_host_cpu.py::test_ablation_patch_applies_to_its_base checks that patch and base still match.
set -e
ABL_BASE=7530589
root="$(cd "$(dirname "$0")/../.." && pwd)"
cd "$root/rumpy_amd/csrc"
src=$1; macro=$2; shift 2
make -s -j6
scratch="$root/build_abl/_src"; rm -rf "$scratch"; mkdir -p "$scratch"
git -C "$root" archive $ABL_BASE rumpy_amd/csrc include | tar -x -C "$scratch"
(cd "$scratch" && git init -q . 2>/dev/null; git -C "$scratch" apply "$root/tests/tools/patches/abl_r03.patch")
for v in "$@"; do
  d=../../build_abl/${macro}_$v$TAG; mkdir -p $d
  src_dir="$scratch/rumpy_amd/csrc"
  (cd "$src_dir" && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $( [ "$src" = conv_rcab.hip ] && echo -Xclang -target-feature -Xclang -packed-fp32-ops ) -D${macro}=$v $EXTRA -c $src -o "$root/build_abl/${macro}_$v$TAG/abl.o")
  objs=$(ls *.o | grep -v "^${src%.hip}.o$")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/librumpy_amd.so $objs $d/abl.o
  rm $d/abl.o
done
