cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_x; rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_chain_gpu.py tests/test_network_gpu.py -m gpu -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
bash tests/tools/pmc_step.sh edsr > $O/pmc_step_edsr.txt 2>&1; cp gpurun_out/pmc_traffic_edsr.json $O/; tail -3 $O/pmc_step_edsr.txt
python3 bench.py --no-cpu-baseline > $O/bench_line.json 2>> $O/err.log; tail -c 600 $O/bench_line.json
