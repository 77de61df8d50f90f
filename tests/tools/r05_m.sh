cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_m; rm -rf $O; mkdir -p $O
V=$GRAFT_REPO_ROOT/build_abl/CHAIN_EARLY/librumpy_amd.so
RUMPY_AMD_LIB=$V python3 -m pytest tests/test_chain_gpu.py -x -q -k "chain_is_bitwise and not rcab" > $O/tests.log 2>&1; tail -3 $O/tests.log
RUMPY_AMD_LIB=$GRAFT_REPO_ROOT/build_abl/CHAIN_EARLY_STAMPS/librumpy_amd.so python3 tests/tools/chain_stamps.py 2>&1 | grep -v amdgpu.ids | tail -8
for rep in 1 2 3; do
  python3 bench.py --no-cpu-baseline --settled-probe-ms 0 > $O/edsr_late_$rep.json 2>> $O/err.log
  RUMPY_AMD_LIB=$V python3 bench.py --no-cpu-baseline --settled-probe-ms 0 > $O/edsr_early_$rep.json 2>> $O/err.log
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
