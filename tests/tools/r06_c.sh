cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_c; rm -rf $O; mkdir -p $O
timeout 600 python3 -m pytest tests/test_chain_gpu.py -x -q -k "one_wave or form1" > $O/chain_tests.log 2>&1; tail -3 $O/chain_tests.log
timeout 300 python3 tests/tools/chain_forms_time.py 16 32 30 > $O/forms_time.txt 2>&1; grep "us per" $O/forms_time.txt
for v in $VARIANTS; do RUMPY_AMD_LIB=build_abl/$v/librumpy_amd.so timeout 300 python3 tests/tools/chain_forms_time.py 16 32 30 > $O/forms_time_$v.txt 2>&1; echo $v; grep "chain1.*us per" $O/forms_time_$v.txt; done
if [ -n "$STAMPS" ]; then C1_STAMPS=1 RUMPY_AMD_LIB=build_abl/C1_STAMPS/librumpy_amd.so timeout 300 python3 tests/tools/chain_forms_time.py 16 32 10 > $O/stamps.txt 2>&1; grep -A15 "phase durations" $O/stamps.txt; fi
