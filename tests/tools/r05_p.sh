# round 5: full GPU suite on the begin-less chain + fused tail gradient, then the headline three times
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_p; rm -rf $O; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -q -x > $O/gpu_suite.log 2>&1; tail -4 $O/gpu_suite.log
for i in 1 2 3; do python3 bench.py --steps 400 --warmup 60 --no-cpu-baseline > $O/edsr_$i.json 2>> $O/err.log; done
python3 bench.py --model rcan --no-cpu-baseline > $O/rcan.json 2>> $O/err.log
python3 - <<PY
import json, glob
for f in sorted(glob.glob('$O/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_us'))
    except Exception as e:
        print(f, 'NO LINE', e)
PY
tail -3 $O/err.log
