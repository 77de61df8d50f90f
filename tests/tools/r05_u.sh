# round 5: the body-end conv (and its data gradient) inside the chain launch: parity, then the step A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_u; rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_chain_gpu.py -m gpu -q -x > $O/tests.log 2>&1; grep -v "^  \|^$" $O/tests.log | tail -15
for i in 1 2 3; do
  RUMPY_NO_CHAIN_EDGE=1 python3 bench.py --steps 400 --warmup 60 --no-cpu-baseline --no-as-called > $O/edsr_sep_$i.json 2>> $O/err.log
  python3 bench.py --steps 400 --warmup 60 --no-cpu-baseline --no-as-called > $O/edsr_edge_$i.json 2>> $O/err.log
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
