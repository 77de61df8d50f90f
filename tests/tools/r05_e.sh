cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_e; rm -rf $O; mkdir -p $O
OLD=$GRAFT_REPO_ROOT/build_abl/R4_RCAB/librumpy_amd.so
python3 -m pytest tests/test_rcab2_gpu.py tests/test_network_gpu.py -x -q -k "rcab2 or one_launch_rcab or rcan_full_depth or rcan_small" > $O/tests.log 2>&1; tail -4 $O/tests.log
python3 tests/tools/kbench.py rcab > $O/kbench.txt 2>&1; RUMPY_AMD_LIB=$OLD python3 tests/tools/kbench.py rcab >> $O/kbench.txt 2>&1; python3 tests/tools/kbench.py rcab2 >> $O/kbench.txt 2>&1; grep "us per launch" $O/kbench.txt
for rep in 1 2; do
  RUMPY_RCAB_FORM=xchg python3 bench.py --model rcan --no-cpu-baseline --steps 100 --warmup 20 --settled-probe-ms 0 > $O/rcan_xchg_new_$rep.json 2>> $O/err.log
  RUMPY_RCAB_FORM=xchg RUMPY_AMD_LIB=$OLD python3 bench.py --model rcan --no-cpu-baseline --steps 100 --warmup 20 --settled-probe-ms 0 > $O/rcan_xchg_r4_$rep.json 2>> $O/err.log
  RUMPY_RCAB_FORM=lazy python3 bench.py --model rcan --no-cpu-baseline --steps 100 --warmup 20 --settled-probe-ms 0 > $O/rcan_lazy_$rep.json 2>> $O/err.log
  RUMPY_RCAB_FORM=xchg python3 bench.py --model rcan --precision fp8 --no-cpu-baseline --steps 100 --warmup 20 --settled-probe-ms 0 > $O/rcan_fp8_new_$rep.json 2>> $O/err.log
  RUMPY_RCAB_FORM=xchg RUMPY_AMD_LIB=$OLD python3 bench.py --model rcan --precision fp8 --no-cpu-baseline --steps 100 --warmup 20 --settled-probe-ms 0 > $O/rcan_fp8_r4_$rep.json 2>> $O/err.log
done
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob('$O/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('avg_launch_us'), d['config']['loss'])
    except Exception as e:
        print(f, 'NO LINE', e)
PY
tail -3 $O/err.log
