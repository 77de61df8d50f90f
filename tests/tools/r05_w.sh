# round 5: fragment reads two groups ahead in the chain's two-row sweeps (build_abl/chain_ahead2) against one group ahead
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_w; rm -rf $O; mkdir -p $O
RUMPY_AMD_LIB=$GRAFT_REPO_ROOT/build_abl/chain_ahead2/librumpy_amd.so timeout 600 python3 -m pytest tests/test_chain_gpu.py -m gpu -q -x -k "bitwise" > $O/tests.log 2>&1; tail -2 $O/tests.log
for i in 1 2 3; do
  python3 bench.py --steps 400 --warmup 60 --no-cpu-baseline --no-as-called > $O/edsr_a1_$i.json 2>> $O/err.log
  RUMPY_AMD_LIB=$GRAFT_REPO_ROOT/build_abl/chain_ahead2/librumpy_amd.so python3 bench.py --steps 400 --warmup 60 --no-cpu-baseline --no-as-called > $O/edsr_a2_$i.json 2>> $O/err.log
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob('$O/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_us'), d['roofline'].get('traffic'))
    except Exception as e:
        print(f, 'NO LINE', e)
PY
tail -3 $O/err.log
