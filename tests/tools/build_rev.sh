#!/bin/bash
# Same-box A/B against an earlier revision: builds rumpy_amd/csrc of git revision <rev> into build_abl/rev_<name>/librumpy_amd.so
# usage: tests/tools/build_rev.sh <rev> [name]      (select with RUMPY_AMD_LIB=build_abl/rev_<name>/librumpy_amd.so; the C ABI must match)
set -e
cd "$(dirname "$0")/../.."
rev=$1; name=${2:-$1}
d=build_abl/rev_$name; rm -rf $d; mkdir -p $d/src/rumpy_amd $d/src/include
git archive $rev rumpy_amd/csrc include | tar -x -C $d/src
make -s -C $d/src/rumpy_amd/csrc -j8 OUT=../../../librumpy_amd.so
rm -rf $d/src
ls -la $d/librumpy_amd.so
