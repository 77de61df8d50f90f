# round 6: wgrad_dma_kernel with a unit's fragment reads pinned between its MFMAs (build_abl/WGRAD_PIN1) against the product build, same box
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_f; rm -rf $O; mkdir -p $O
timeout 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "wgrad" > $O/wgrad_tests.log 2>&1; tail -2 $O/wgrad_tests.log
RUMPY_AMD_LIB=build_abl/WGRAD_PIN1/librumpy_amd.so timeout 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "wgrad" > $O/wgrad_tests_pin.log 2>&1; tail -2 $O/wgrad_tests_pin.log
for i in 1 2 3; do
  python3 bench.py --no-as-called > $O/bench_base_$i.json 2>/dev/null; python3 -c "import json;d=json.load(open('$O/bench_base_$i.json'));print('base', d['value'], d['settled']['value'])"
  RUMPY_AMD_LIB=build_abl/WGRAD_PIN1/librumpy_amd.so python3 bench.py --no-as-called > $O/bench_pin_$i.json 2>/dev/null; python3 -c "import json;d=json.load(open('$O/bench_pin_$i.json'));print('pin ', d['value'], d['settled']['value'])"
done
cd /tmp && export TMPDIR=/tmp
for v in base pin; do
  if [ $v = pin ]; then export RUMPY_AMD_LIB=$GRAFT_REPO_ROOT/build_abl/WGRAD_PIN1/librumpy_amd.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$v -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-as-called --settled-probe-ms 0 > $GRAFT_REPO_ROOT/$O/prof_$v.log 2>&1
  grep -h "wgrad_dma" $(find $GRAFT_REPO_ROOT/$O/prof_$v -name '*kernel_stats.csv' | head -1) | cut -c1-120
done
