# round 6: conv4d / conv4dt with a unit's LDS reads pinned between its MFMAs (build_abl/D4_PIN1) against the product build (wgrad pin + block pin shipped), same box
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_i; rm -rf $O; mkdir -p $O
RUMPY_AMD_LIB=build_abl/D4_PIN1/librumpy_amd.so timeout 900 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "upsampler or conv4 or tail or dgrad" > $O/tests_pin.log 2>&1; tail -2 $O/tests_pin.log
for i in 1 2 3; do
  python3 bench.py --no-as-called > $O/b.json 2>/dev/null; python3 -c "import json;d=json.load(open('$O/b.json'));print('edsr base', d['value'], d['settled']['value'])"
  RUMPY_AMD_LIB=build_abl/D4_PIN1/librumpy_amd.so python3 bench.py --no-as-called > $O/b.json 2>/dev/null; python3 -c "import json;d=json.load(open('$O/b.json'));print('edsr pin ', d['value'], d['settled']['value'])"
done
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for v in base D4_PIN1; do
  if [ $v = base ]; then unset RUMPY_AMD_LIB; else export RUMPY_AMD_LIB=$R/build_abl/$v/librumpy_amd.so; fi
  rm -rf $R/$O/prof; rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o p -- python3 $R/bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-as-called --settled-probe-ms 0 > $R/$O/prof_$v.log 2>&1
  echo "== $v"; python3 $R/tests/tools/prof_summary.py $(find $R/$O/prof -name '*kernel_stats.csv' | head -1) 1 | sort -k1,1 -n -r | head -8
done
