# round 5: s_setprio around the chain's sweeps (build_abl/chain_prio1 / chain_prio3) against none
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_y; rm -rf $O; mkdir -p $O
for i in 1 2 3; do
  python3 bench.py --steps 400 --warmup 60 --no-cpu-baseline --no-as-called > $O/edsr_p0_$i.json 2>> $O/err.log
  for pr in 1 3; do
    RUMPY_AMD_LIB=$GRAFT_REPO_ROOT/build_abl/chain_prio$pr/librumpy_amd.so python3 bench.py --steps 400 --warmup 60 --no-cpu-baseline --no-as-called > $O/edsr_p${pr}_$i.json 2>> $O/err.log
  done
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
