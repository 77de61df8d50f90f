# round 6: block-record pointers of conv_chain.hip through the global address space (no flat_* instructions) against the flat build (build_abl/CHAIN_FLAT), same box:
# chain tests, isolated chain times, step A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_e; rm -rf $O; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_chain_gpu.py -x -q > $O/chain_tests.log 2>&1; tail -4 $O/chain_tests.log
timeout 600 python3 -m pytest tests/tools/chain_tests.py -m tools -x -q -k "one_wave or rcab_chain" > $O/tools_tests.log 2>&1; tail -2 $O/tools_tests.log
for i in 1 2; do
timeout 300 python3 tests/tools/chain_forms_time.py 16 32 30 2>&1 | grep "res_chain .*us per" | sed 's/^/global  /'
RUMPY_AMD_LIB=build_abl/CHAIN_FLAT/librumpy_amd.so timeout 300 python3 tests/tools/chain_forms_time.py 16 32 30 2>&1 | grep "res_chain .*us per" | sed 's/^/flat    /'
done
for i in 1 2 3; do
  python3 bench.py --no-as-called > $O/bench_global_$i.json 2>/dev/null; python3 -c "import json;d=json.load(open('$O/bench_global_$i.json'));print('global', d['value'], d['settled']['value'])"
  RUMPY_AMD_LIB=build_abl/CHAIN_FLAT/librumpy_amd.so python3 bench.py --no-as-called > $O/bench_flat_$i.json 2>/dev/null; python3 -c "import json;d=json.load(open('$O/bench_flat_$i.json'));print('flat  ', d['value'], d['settled']['value'])"
done
