cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_j; rm -rf $O; mkdir -p $O
python3 -m pytest tests/test_chain_gpu.py -x -q -k "rcab_chain or rcan_training" > $O/tests.log 2>&1; tail -5 $O/tests.log
for rep in 1 2; do
  python3 bench.py --model rcan --no-cpu-baseline --steps 100 --warmup 20 --settled-probe-ms 0 > $O/rcan_chain_$rep.json 2>> $O/err.log
  RUMPY_NO_CHAIN=1 python3 bench.py --model rcan --no-cpu-baseline --steps 100 --warmup 20 --settled-probe-ms 0 > $O/rcan_blocks_$rep.json 2>> $O/err.log
  RUMPY_CHAIN_SC1=1 python3 bench.py --model rcan --no-cpu-baseline --steps 100 --warmup 20 --settled-probe-ms 0 > $O/rcan_chainsc1_$rep.json 2>> $O/err.log
done
python3 bench.py --model blindqrcan --no-cpu-baseline --steps 100 --warmup 20 --settled-probe-ms 0 > $O/blind_chain.json 2>> $O/err.log
RUMPY_NO_CHAIN=1 python3 bench.py --model blindqrcan --no-cpu-baseline --steps 100 --warmup 20 --settled-probe-ms 0 > $O/blind_blocks.json 2>> $O/err.log
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
tail -3 $O/err.log
