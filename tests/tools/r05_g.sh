cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_g; rm -rf $O; mkdir -p $O
python3 tests/tools/as_called_prof.py > $O/as_called_prof.txt 2>&1; grep "ms per call" $O/as_called_prof.txt
bash tests/tools/r05_chain.sh > $O/chain.txt 2>&1; cat $O/chain.txt
python3 tests/tools/r05_fp8_joint_diag.py > $O/fp8_joint_diag.txt 2>&1; grep "fused-L1" $O/fp8_joint_diag.txt
python3 -m pytest tests/test_fp8_gpu.py -q -k "config5 or mean_reduced" > $O/fp8_pin.log 2>&1; grep "config 5\|first-pass\|passed\|failed" $O/fp8_pin.log
