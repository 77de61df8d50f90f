# round 5: RCAB forward tail (one barrier less in the pool all-gather, t2 stores in front of the MLP) against the previous build (build_abl/rcab_old)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_z; rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_network_gpu.py tests/test_fp8_gpu.py tests/test_qrcan_gpu.py -m gpu -q -x -k "rcab or rcan or fp8" > $O/tests.log 2>&1; tail -3 $O/tests.log
for i in 1 2 3; do
  RUMPY_AMD_LIB=$GRAFT_REPO_ROOT/build_abl/rcab_old/librumpy_amd.so python3 bench.py --model rcan --steps 100 --warmup 20 --no-cpu-baseline --no-as-called > $O/rcan_old_$i.json 2>> $O/err.log
  python3 bench.py --model rcan --steps 100 --warmup 20 --no-cpu-baseline --no-as-called > $O/rcan_new_$i.json 2>> $O/err.log
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
