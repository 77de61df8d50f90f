cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_c; rm -rf $O; mkdir -p $O
python3 -m pytest tests/test_rcab2_gpu.py -x -q > $O/kernel_tests.log 2>&1; tail -4 $O/kernel_tests.log
bash tests/tools/r05_stamps.sh run > $O/stamps.txt 2>&1; grep -v amdgpu.ids $O/stamps.txt
python3 tests/tools/kbench.py rcab > $O/kbench.txt 2>&1; python3 tests/tools/kbench.py rcab2 >> $O/kbench.txt 2>&1; grep "us per launch" $O/kbench.txt
python3 tests/tools/host_stage_time.py > $O/host_stage.txt 2>&1; grep ms $O/host_stage.txt
for rep in 1 2; do for form in lazy xchg; do
  RUMPY_RCAB_FORM=$form python3 bench.py --model rcan --no-cpu-baseline --steps 100 --warmup 20 --settled-probe-ms 0 > $O/rcan_${form}_$rep.json 2>> $O/err.log
done; done
python3 - <<PY
import json
for rep in (1, 2):
    for form in ('lazy', 'xchg'):
        try:
            d = json.loads(open('$O/rcan_%s_%d.json' % (form, rep)).read().strip().splitlines()[-1])
            print(form, rep, d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('avg_launch_us'))
        except Exception as e:
            print(form, rep, 'NO LINE', e)
PY
