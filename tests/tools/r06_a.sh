# round 6: watchdog fall-back tests, the moved RCAB-chain kernel's parity test, bench line (the polls' status look-up must cost nothing)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_a; rm -rf $O; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_chain_gpu.py -x -q > $O/chain_tests.log 2>&1; tail -8 $O/chain_tests.log
timeout 600 python3 -m pytest tests/tools/chain_tests.py -m tools -x -q -k "rcab_chain" > $O/tools_tests.log 2>&1; tail -3 $O/tools_tests.log
timeout 600 python3 -m pytest tests/test_fp8_gpu.py -x -q -k "config5" -rx > $O/fp8_tests.log 2>&1; tail -6 $O/fp8_tests.log
python3 bench.py --no-as-called > $O/bench.json 2> $O/bench.err; cat $O/bench.json | cut -c1-400
