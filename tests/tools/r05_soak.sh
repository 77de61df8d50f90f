# round 5: soak - 3 x 6000 EDSR steps on the chain (hand-off watchdog and loss must stay clean), then 1500 RCAN steps
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_soak; rm -rf $O; mkdir -p $O
for i in 1 2 3; do python3 bench.py --steps 6000 --warmup 50 --no-cpu-baseline --no-as-called --settled-probe-ms 0 > $O/edsr_$i.json 2>> $O/err.log; done
python3 bench.py --model rcan --steps 1500 --warmup 20 --no-cpu-baseline --no-as-called --settled-probe-ms 0 > $O/rcan.json 2>> $O/err.log
python3 - <<PY
import json, glob
for f in sorted(glob.glob('$O/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['steps'], d['config'].get('loss'))
    except Exception as e:
        print(f, 'NO LINE', e)
PY
tail -5 $O/err.log
