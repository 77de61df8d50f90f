cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05_b
bash tests/tools/r05_stamps.sh run > gpurun_out/r05_b/stamps.txt 2>&1; cat gpurun_out/r05_b/stamps.txt | grep -v amdgpu.ids
python3 -m pytest tests/test_network_gpu.py -x -q -k "host_tensors or bench_line_value" > gpurun_out/r05_b/tests.log 2>&1; tail -5 gpurun_out/r05_b/tests.log
python3 bench.py --no-cpu-baseline > gpurun_out/r05_b/edsr_line.json 2> gpurun_out/r05_b/err.log
RUMPY_HOST_STAGING=0 python3 bench.py --no-cpu-baseline --settled-probe-ms 0 > gpurun_out/r05_b/edsr_line_nostage.json 2>> gpurun_out/r05_b/err.log
python3 - <<PY
import json
for f in ('edsr_line.json','edsr_line_nostage.json'):
    d = json.loads(open('gpurun_out/r05_b/'+f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d.get('settled'), d.get('as_called'))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05_b/copytrace -o p -- python3 $GRAFT_REPO_ROOT/bench.py --model rcan --steps 20 --warmup 5 --probe-steps 1 --no-cpu-baseline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 tests/tools/copy_attrib.py $(find gpurun_out/r05_b/copytrace -name '*kernel_trace.csv' | head -1) 20 5 | tee gpurun_out/r05_b/copy_attrib.txt
rm -rf gpurun_out/r05_b/copytrace
