#!/bin/bash
# round 3, upsampler path + batch-by-pointer: bash tests/tools/r03_up.sh  (GPU box; results under gpurun_out/r03_up/)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_up; mkdir -p $O; cd $R
python -m pytest tests/test_network_gpu.py -q -x -k "pointer or hipgraph" > $O/pytest_ptr.txt 2>&1; tail -3 $O/pytest_ptr.txt
python -m pytest tests/test_kernels_gpu.py -q -x -k "conv4 or cin256 or dgrad or 256" > $O/pytest_conv4.txt 2>&1; tail -3 $O/pytest_conv4.txt
for i in 1 2; do
  echo "== conv4 base (XCD order)"; python tests/tools/kbench.py conv4 2>&1 | grep -v amdgpu.ids
  echo "== conv4 D4_XCD_0"; RUMPY_AMD_LIB=$R/build_abl/D4_XCD_0/librumpy_amd.so python tests/tools/kbench.py conv4 2>&1 | grep -v amdgpu.ids
done > $O/conv4_ab.txt 2>&1
cat $O/conv4_ab.txt
echo "== rcab stamps"; RUMPY_AMD_LIB=$R/build_abl/RCAB_ABL_9/librumpy_amd.so python tests/tools/kbench.py rcab > $O/rcab_stamps.txt 2>&1; grep -v amdgpu.ids $O/rcab_stamps.txt
for v in "" "RUMPY_GRAPH=1" "RUMPY_GRAPH=1 RUMPY_BATCH_COPY=1" "" "RUMPY_GRAPH=1"; do
  echo "== bench $v"; env $v python bench.py --steps 300 --warmup 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
done > $O/bench_graph_ab.txt 2>&1
cat $O/bench_graph_ab.txt
