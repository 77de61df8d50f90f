cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_d; rm -rf $O; mkdir -p $O
for rep in 1 2; do for form in lazy xchg; do
  RUMPY_RCAB_FORM=$form python3 bench.py --mode eval --model rcan --steps 20 --warmup 3 > $O/eval_rcan_${form}_$rep.json 2>> $O/err.log
  RUMPY_RCAB_FORM=$form python3 bench.py --model rcan --lr-size 64 --batch 8 --steps 40 --warmup 8 --no-cpu-baseline --settled-probe-ms 0 > $O/rcan64_${form}_$rep.json 2>> $O/err.log
done; done
RUMPY_RCAB_FORM=lazy python3 bench.py --model rcan --lr-size 96 --batch 8 --steps 20 --warmup 4 --no-cpu-baseline --settled-probe-ms 0 > $O/rcan96_lazy.json 2>> $O/err.log
RUMPY_RCAB_FORM=xchg python3 bench.py --model rcan --lr-size 96 --batch 8 --steps 20 --warmup 4 --no-cpu-baseline --settled-probe-ms 0 > $O/rcan96_xchg.json 2>> $O/err.log
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob('$O/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d['value'], d['unit'], d['ms_per_step'], (d.get('roofline') or {}).get('avg_launch_us'), (d.get('roofline') or {}).get('kernel', '')[:40])
    except Exception as e:
        print(f, 'NO LINE', e)
PY
tail -3 $O/err.log
