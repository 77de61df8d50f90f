# round 6: step-level A/B of the chain forms (RUMPY_CHAIN_FORM=2: conv_chain.hip, two waves per SIMD, body-end conv inside; =1: conv_chain1.hip, one wave per SIMD),
# alternating on one box; the chain / watchdog tests (time-based watchdog)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_d; rm -rf $O; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_chain_gpu.py -x -q > $O/chain_tests.log 2>&1; tail -5 $O/chain_tests.log
for i in 1 2 3; do
  for f in 2 1; do RUMPY_CHAIN_FORM=$f python3 bench.py --no-as-called > $O/bench_form${f}_$i.json 2>/dev/null; python3 -c "import json;d=json.load(open('$O/bench_form${f}_$i.json'));print('form $f', d['value'], d['settled']['value'], d['roofline'].get('kernel_us'))"; done
done
RUMPY_CHAIN_FORM=2 RUMPY_NO_CHAIN_EDGE=1 python3 bench.py --no-as-called > $O/bench_form2_noedge.json 2>/dev/null; python3 -c "import json;d=json.load(open('$O/bench_form2_noedge.json'));print('form 2, body-end conv as its own launch', d['value'], d['settled']['value'])"
