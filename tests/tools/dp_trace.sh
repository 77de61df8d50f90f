#!/bin/bash
# kernel timeline of one data-parallel step on ONE rank over RCCL (what sits between the weight gradients and the optimizer launch)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/dptrace; rm -rf $OUT; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0 RUMPY_DP_FORCE=1
cd /tmp && export TMPDIR=/tmp
# the rank itself stands behind `--` (no launcher in between: the profiler's preloaded library may already have initialised the GPU in the
# process it starts, and a launcher's fork + exec from there is what this pool forbids); bench.py takes the rendezvous from these variables
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29599
rocprofv3 --kernel-trace --output-format csv -d $OUT -o p -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 10 --probe-steps 0 --no-cpu-baseline > $OUT/log 2>&1
cd $R
python3 - <<PY
import csv, glob
f = sorted(glob.glob('$OUT/**/*kernel_trace.csv', recursive=True), key=lambda p: -__import__('os').path.getsize(p))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam' in r['Kernel_Name']]
i1 = idx[-3]
i0 = max(j for j in range(i1) if 'conv_block' in rows[j]['Kernel_Name'])
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i1 + 2]:
    print('%9.1f us  +%7.1f us  q%-3s %s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Queue_Id'], r['Kernel_Name'][:70]))
PY
