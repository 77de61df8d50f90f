cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_t; rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_wide_gpu.py tests/test_embedded_widths_gpu.py -m gpu -q > $O/wide.log 2>&1; grep -v "^  \|^$" $O/wide.log | tail -30
timeout 1500 python3 -m pytest tests -m gpu -q -x > $O/gpu_suite.log 2>&1; tail -4 $O/gpu_suite.log
