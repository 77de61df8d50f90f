# round 5: three-row chain with the upper half of the batch started late (anti-phase pairs per CU?)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_r; rm -rf $O; mkdir -p $O
python3 bench.py --steps 400 --warmup 60 --no-cpu-baseline --no-as-called > $O/edsr_geo6.json 2>> $O/err.log
for sk in 0 200 400 600 800 1200; do
  RUMPY_CHAIN_GEO=3 RUMPY_CHAIN_SKEW=$sk python3 bench.py --steps 400 --warmup 60 --no-cpu-baseline --no-as-called > $O/edsr_geo3_skew$sk.json 2>> $O/err.log
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob('$O/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_us'))
    except Exception as e:
        print(f, 'NO LINE', e)
PY
tail -3 $O/err.log
