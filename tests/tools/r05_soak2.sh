cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_soak2; rm -rf $O; mkdir -p $O
for k in 400 6000 400 20000 400; do python3 bench.py --steps $k --warmup 50 --no-cpu-baseline --no-as-called --settled-probe-ms 0 > $O/edsr_$k_$RANDOM.json 2>> $O/err.log; tail -c 4000 $(ls -t $O/*.json | head -1) | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['steps'], d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"; done
rocm-smi --showclocks --showpower 2>/dev/null | head -20
