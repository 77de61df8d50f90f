#!/bin/bash
# round 3: encoder-training parity against the fp32 oracle with fp16 forward storage (and the all-bf16 forward pass for comparison)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r03_enc}; mkdir -p $OUT; cd $R
python -m pytest tests/test_contrastive_gpu.py tests/test_encoder_gpu.py tests/test_blind_gpu.py -q -s -x > $OUT/t_fp16.log 2>&1; echo "fp16 rc=$?" | tee $OUT/summary
grep -h "gradient vs fp32\|worst trunk" $OUT/t_fp16.log
tail -n 3 $OUT/t_fp16.log
RUMPY_ENC_TRAIN_BF16=1 python -m pytest tests/test_contrastive_gpu.py -q -s -k "trunk_forward or moco_training_step or supmoco_training or weakcon_training or supcon_training" > $OUT/t_bf16.log 2>&1; echo "bf16 (expected to miss the new bounds) rc=$?" | tee -a $OUT/summary
grep -h "gradient vs fp32\|AssertionError\|assert whole\|^E  " $OUT/t_bf16.log | head -30
python bench.py --model moco --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | grep '^{"metric"' > $OUT/moco_fp16.json
RUMPY_ENC_TRAIN_BF16=1 python bench.py --model moco --steps 200 --warmup 20 --no-cpu-baseline 2>&1 | grep '^{"metric"' > $OUT/moco_bf16.json
python -c "
import json
for f in ('moco_fp16','moco_bf16'):
    d=json.loads(open('$OUT/'+f+'.json').read()); print(f, d['value'], d['unit'], d['ms_per_step'], 'ms')
"
