# round 6, VERDICT r5 item 1: per-block ablation of the fp8 joint-loss cases (tests/tools/fp8_block_ablation.py)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06_fp8; mkdir -p $O
timeout 500 python3 tests/tools/fp8_block_ablation.py supmoco --wq --policies 'every:10' --budget 400 > $O/supmoco_wq.txt 2>&1
timeout 900 python3 tests/tools/fp8_block_ablation.py supmoco --policies 'every:10;every:20' --random 12 --singles --budget 800 > $O/supmoco_singles.txt 2>&1
timeout 300 python3 tests/tools/fp8_block_ablation.py moco --budget 250 > $O/moco.txt 2>&1
timeout 300 python3 tests/tools/fp8_block_ablation.py frozen --budget 250 > $O/frozen.txt 2>&1
grep -h "whole\|#" $O/supmoco_wq.txt | tail -5; tail -3 $O/supmoco_singles.txt; grep -h "whole" $O/moco.txt | head -30
