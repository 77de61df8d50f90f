cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_suite; rm -rf $O; mkdir -p $O
python3 -m pytest tests -m gpu -q -x -rx > $O/gpu_suite.log 2>&1; tail -12 $O/gpu_suite.log | cut -c1-300
python3 -m pytest tests/tools/chain_tests.py -m tools -q -x > $O/tools_suite.log 2>&1; tail -2 $O/tools_suite.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
