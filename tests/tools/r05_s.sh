cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_s; rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_qrcan_gpu.py tests/test_blind_gpu.py -m gpu -q -x > $O/q.log 2>&1; tail -6 $O/q.log
timeout 1500 python3 -m pytest tests -m gpu -q -x > $O/gpu_suite.log 2>&1; tail -4 $O/gpu_suite.log
