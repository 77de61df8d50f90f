#!/bin/bash
# round 3: the contrastive head in HIP - tests, the MoCo step's kernel list (eager step: every launch visible to the profiler), bench line
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r03_head}; mkdir -p $OUT; cd $R
python -m pytest tests/test_contrastive_gpu.py tests/test_blind_gpu.py -q -s > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee $OUT/summary
grep -h "gradient vs fp32\|worst trunk\|^FAILED\|passed\|failed" $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
RUMPY_MOCO_STEP_GRAPH=0 RUMPY_ENC_GRAPH=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python3 $R/tests/tools/moco_time.py 32 48 60 > $OUT/moco_time_eager.log 2>&1
cd $R
python tests/tools/prof_summary.py $(find $OUT/prof -name '*kernel_stats.csv' | head -1) 70 > $OUT/moco_step_kernels.txt 2>&1
head -45 $OUT/moco_step_kernels.txt
python tests/tools/moco_time.py 32 48 200 | tail -1
python tests/tools/moco_time.py 256 48 50 | tail -1
python bench.py --model moco --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | grep '^{"metric"' > $OUT/moco_bench_line.json; python -c "
import json; d=json.loads(open('$OUT/moco_bench_line.json').read()); print(d['value'], d['unit'], d['ms_per_step'])"
