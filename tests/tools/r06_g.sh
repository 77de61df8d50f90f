# round 6: wgrad_dma_kernel variants (pin pattern, read-ahead depth), kernel time from rocprofv3 --kernel-trace --stats of a 60-step bench run each
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_g; rm -rf $O; mkdir -p $O
for v in base WGRAD_PIN1 $VARIANTS base WGRAD_PIN1; do
  if [ $v = base ]; then unset RUMPY_AMD_LIB; else export RUMPY_AMD_LIB=$R/build_abl/$v/librumpy_amd.so; fi
  rm -rf $O/prof; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 $R/bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-as-called --settled-probe-ms 0 > $O/prof_$v.log 2>&1
  echo "$v $(grep -h 'wgrad_dma' $(find $O/prof -name '*kernel_stats.csv' | head -1) | awk -F, '{print $(NF-4)}') $(python3 -c "import json;print(json.loads(open('$O/prof_$v.log').read().strip().split(chr(10))[-1])['value'])")"
done
