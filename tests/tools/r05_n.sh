# round 5: the tail conv's data gradient inside the last upsampler stage's data-gradient launch (rumpy_conv4d_tail): parity, kernel timing, step A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_n; rm -rf $O; mkdir -p $O
timeout 600 python3 -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "tail or conv4d" > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 300 python3 tests/tools/kbench.py tailfuse > $O/kbench.log 2>&1; cat $O/kbench.log
for i in 1 2; do
  RUMPY_NO_TAIL_FUSE=1 python3 bench.py --steps 400 --warmup 60 --no-cpu-baseline > $O/edsr_sep_$i.json 2>> $O/err.log
  python3 bench.py --steps 400 --warmup 60 --no-cpu-baseline > $O/edsr_fused_$i.json 2>> $O/err.log
done
RUMPY_NO_TAIL_FUSE=1 python3 bench.py --model rcan --no-cpu-baseline > $O/rcan_sep.json 2>> $O/err.log
python3 bench.py --model rcan --no-cpu-baseline > $O/rcan_fused.json 2>> $O/err.log
python3 - <<PY
import json, glob
for f in sorted(glob.glob('$O/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'])
    except Exception as e:
        print(f, 'NO LINE', e)
PY
tail -5 $O/err.log
timeout 900 python3 -m pytest tests/test_network_gpu.py tests/test_chain_gpu.py -m gpu -q -x > $O/net_tests.log 2>&1; tail -5 $O/net_tests.log
