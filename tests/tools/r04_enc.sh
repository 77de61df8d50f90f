#!/bin/bash
# encoder training parity (fp32 conv outputs + unrounded filters) and what it costs: tests, MoCo bench with and without.  usage: bash tests/tools/r04_enc.sh <tag>
TAG=${1:-a}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/enc_$TAG; rm -rf $OUT; mkdir -p $OUT; cd $R
timeout 1200 python3 -m pytest tests/test_contrastive_gpu.py tests/test_encoder_gpu.py tests/test_blind_gpu.py -q -s -x > $OUT/pytest.log 2>&1; echo "rc $?" >> $OUT/pytest.log
grep -n "gradient vs fp32 oracle\|worst trunk\|passed\|failed\|Error\|rc " $OUT/pytest.log | tail -30
python3 bench.py --model moco --steps 300 --warmup 30 --no-cpu-baseline > $OUT/moco_z32.json 2>> $OUT/err.log
RUMPY_ENC_TRAIN_Z16=1 python3 bench.py --model moco --steps 300 --warmup 30 --no-cpu-baseline > $OUT/moco_z16.json 2>> $OUT/err.log
python3 bench.py --model moco --steps 300 --warmup 30 --no-cpu-baseline > $OUT/moco_z32_2.json 2>> $OUT/err.log
for f in $OUT/moco_*.json; do python3 -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], d['value'], d['unit'], d['ms_per_step'])"; done
tail -3 $OUT/err.log
