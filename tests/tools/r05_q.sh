# round 5: three-row strips, two workgroups per CU (RUMPY_CHAIN_GEO=3): parity, then the step A/B on one box (+ the begin-kernel A/B)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_q; rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_chain_gpu.py -m gpu -q -x -k "three_row" > $O/tests3.log 2>&1; tail -4 $O/tests3.log
for i in 1 2 3; do
  python3 bench.py --steps 400 --warmup 60 --no-cpu-baseline --no-as-called > $O/edsr_geo6_$i.json 2>> $O/err.log
  RUMPY_CHAIN_GEO=3 python3 bench.py --steps 400 --warmup 60 --no-cpu-baseline --no-as-called > $O/edsr_geo3_$i.json 2>> $O/err.log
  RUMPY_AMD_LIB=$GRAFT_REPO_ROOT/build_abl/chain_begin/librumpy_amd.so python3 bench.py --steps 400 --warmup 60 --no-cpu-baseline --no-as-called > $O/edsr_begin_$i.json 2>> $O/err.log
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
timeout 600 python3 -m pytest tests/test_chain_gpu.py -m gpu -q -x > $O/tests_all.log 2>&1; tail -3 $O/tests_all.log
