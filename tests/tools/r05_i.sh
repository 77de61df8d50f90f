cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_i; rm -rf $O; mkdir -p $O
timeout 600 python3 -m pytest tests/test_chain_gpu.py -x -q -k "rcab_chain" > $O/tests.log 2>&1; tail -25 $O/tests.log
