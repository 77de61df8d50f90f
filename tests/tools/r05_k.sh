cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_k; rm -rf $O; mkdir -p $O
python3 -m pytest tests/test_chain_gpu.py -x -q -k "not rcab and not rcan" > $O/tests.log 2>&1; tail -4 $O/tests.log
for rep in 1 2 3; do
  python3 bench.py --no-cpu-baseline --settled-probe-ms 0 > $O/edsr_chain_$rep.json 2>> $O/err.log
  RUMPY_NO_CHAIN=1 python3 bench.py --no-cpu-baseline --settled-probe-ms 0 > $O/edsr_blocks_$rep.json 2>> $O/err.log
done
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob('$O/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get('roofline') or {}
        print(os.path.basename(f), d['value'], d['ms_per_step'], r.get('avg_launch_us'), r.get('frac'), d['config']['loss'])
    except Exception as e:
        print(f, 'NO LINE', e)
PY
