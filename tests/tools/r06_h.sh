# round 6: block_sweep's fragment reads pinned between the MFMAs (build_abl/BLOCK_PIN1: conv_block, conv_chain, conv_rcab, conv_rcab2, conv_up, conv_dgrad4) against
# the product build, same box, alternating: EDSR and RCAN steps, kernel statistics
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_h; rm -rf $O; mkdir -p $O
RUMPY_AMD_LIB=build_abl/BLOCK_PIN1/librumpy_amd.so timeout 900 python3 -m pytest tests/test_chain_gpu.py tests/test_kernels_gpu.py -x -q > $O/tests_pin.log 2>&1; tail -2 $O/tests_pin.log
for i in 1 2 3; do
  python3 bench.py --no-as-called > $O/b.json 2>/dev/null; python3 -c "import json;d=json.load(open('$O/b.json'));print('edsr base', d['value'], d['settled']['value'])"
  RUMPY_AMD_LIB=build_abl/BLOCK_PIN1/librumpy_amd.so python3 bench.py --no-as-called > $O/b.json 2>/dev/null; python3 -c "import json;d=json.load(open('$O/b.json'));print('edsr pin ', d['value'], d['settled']['value'])"
done
for i in 1 2; do
  python3 bench.py --model rcan --no-as-called --no-cpu-baseline > $O/b.json 2>/dev/null; python3 -c "import json;d=json.load(open('$O/b.json'));print('rcan base', d['value'])"
  RUMPY_AMD_LIB=build_abl/BLOCK_PIN1/librumpy_amd.so python3 bench.py --model rcan --no-as-called --no-cpu-baseline > $O/b.json 2>/dev/null; python3 -c "import json;d=json.load(open('$O/b.json'));print('rcan pin ', d['value'])"
done
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for v in base BLOCK_PIN1; do
  if [ $v = base ]; then unset RUMPY_AMD_LIB; else export RUMPY_AMD_LIB=$R/build_abl/$v/librumpy_amd.so; fi
  rm -rf $R/$O/prof; rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o p -- python3 $R/bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-as-called --settled-probe-ms 0 > $R/$O/prof_$v.log 2>&1
  echo "== $v"; python3 $R/tests/tools/prof_summary.py $(find $R/$O/prof -name '*kernel_stats.csv' | head -1) 1 | sort -k1,1 -n -r | head -8
done
