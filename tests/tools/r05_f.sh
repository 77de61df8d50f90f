cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_f; rm -rf $O; mkdir -p $O
python3 tests/tools/as_called_prof.py > $O/as_called_prof.txt 2>&1; grep -v "^$" $O/as_called_prof.txt | head -70
python3 -m pytest tests -m gpu -x -q > $O/gpu_suite.log 2>&1; tail -8 $O/gpu_suite.log
