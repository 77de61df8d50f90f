# round 6: the one-wave-per-SIMD chain (conv_chain1.hip): parity, isolated time per form, phase stamps, step A/B; watchdog tests again (time-based watchdog)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_b; rm -rf $O; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_chain_gpu.py -x -q > $O/chain_tests.log 2>&1; tail -8 $O/chain_tests.log
timeout 300 python3 tests/tools/chain_forms_time.py 16 32 30 > $O/forms_time.txt 2>&1; cat $O/forms_time.txt
C1_STAMPS=1 RUMPY_AMD_LIB=build_abl/C1_STAMPS/librumpy_amd.so timeout 300 python3 tests/tools/chain_forms_time.py 16 32 10 > $O/stamps.txt 2>&1; grep -A14 "phase durations" $O/stamps.txt
for i in 1 2; do
  for f in 2 1; do RUMPY_CHAIN_FORM=$f python3 bench.py --no-as-called > $O/bench_form${f}_$i.json 2>/dev/null; python3 -c "import json;d=json.load(open('$O/bench_form${f}_$i.json'));print('form $f', d['value'], d.get('settled'))"; done
done
