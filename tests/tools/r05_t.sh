cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_t; rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_wide_gpu.py tests/test_qrcan_gpu.py -m gpu -q -k "larger_image or at_128 or refused" > $O/wide.log 2>&1; grep -v "^  \|^$" $O/wide.log | tail -30
