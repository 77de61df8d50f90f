# round 5: conv4d / conv4dt fragment reads three units ahead (D4_RD=3, build_abl/d4_rd3) against two
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_rd3; rm -rf $O; mkdir -p $O
RUMPY_AMD_LIB=$GRAFT_REPO_ROOT/build_abl/d4_rd3/librumpy_amd.so timeout 600 python3 -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "tail_dgrad_inside or streaming_cin256" > $O/tests.log 2>&1; tail -2 $O/tests.log
for i in 1 2 3; do
  python3 tests/tools/kbench.py tailfuse 2>&1 | grep "32x96x96" | tail -1
  RUMPY_AMD_LIB=$GRAFT_REPO_ROOT/build_abl/d4_rd3/librumpy_amd.so python3 tests/tools/kbench.py tailfuse 2>&1 | grep "32x96x96" | tail -1
done
for i in 1 2 3; do
  python3 bench.py --steps 400 --warmup 60 --no-cpu-baseline --no-as-called > $O/edsr_rd2_$i.json 2>> $O/err.log
  RUMPY_AMD_LIB=$GRAFT_REPO_ROOT/build_abl/d4_rd3/librumpy_amd.so python3 bench.py --steps 400 --warmup 60 --no-cpu-baseline --no-as-called > $O/edsr_rd3_$i.json 2>> $O/err.log
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob('$O/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'])
PY
