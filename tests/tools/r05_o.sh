# round 5: where the fused tail + upsampler data gradient spends its time INSIDE the step (rocprofv3 kernel trace per variant)
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_o; rm -rf $O; mkdir -p $O
timeout 600 python3 -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "tail or conv4d" > $O/tests.log 2>&1; tail -3 $O/tests.log
cd /tmp && export TMPDIR=/tmp
run() {  # name, env assignment
  export RUMPY_NO_TAIL_FUSE=$2 RUMPY_C4T_MODE=$3
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$1 -o p -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --settled-probe-ms 0 > $O/p_$1.log 2>&1
  cp $(find $O/p_$1 -name '*kernel_stats.csv' | head -1) $O/$1_kernel_stats.csv; rm -rf $O/p_$1
  echo "== $1: $(tail -1 $O/p_$1.log | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  python3 $R/tests/tools/prof_summary.py $O/$1_kernel_stats.csv 148 | grep -i "conv4d\|tail_dgrad\|wgrad_dma\|conv_up\|tail_fwd" 
}
run sep 1 0
run fused 0 0
run fused_nodx 0 1
run fused_plain 0 2
