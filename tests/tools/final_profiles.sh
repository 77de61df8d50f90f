#!/bin/bash
# Round artefacts for profiles/: the default bench line per model, and rocprofv3 --kernel-trace --stats of the same command (EDSR, RCAN).
# usage (GPU box): bash tests/tools/final_profiles.sh <tag>      -> gpurun_out/final_<tag>/
TAG=${1:-x}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $R
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench_err.log
python3 bench.py --model rcan > $OUT/rcan_bench_line.json 2>> $OUT/bench_err.log
python3 bench.py --model blindqrcan > $OUT/blindqrcan_bench_line.json 2>> $OUT/bench_err.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_edsr -o p -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline > $OUT/prof_edsr.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_rcan -o p -- python3 $R/bench.py --model rcan --steps 40 --warmup 10 --no-cpu-baseline > $OUT/prof_rcan.log 2>&1
cd $R
cp $OUT/prof_edsr/*kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null || cp $(find $OUT/prof_edsr -name '*kernel_stats.csv' | head -1) $OUT/bench_kernel_stats.csv
cp $(find $OUT/prof_rcan -name '*kernel_stats.csv' | head -1) $OUT/rcan_kernel_stats.csv
rm -rf $OUT/prof_edsr $OUT/prof_rcan
grep -h -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"avg_launch_us": [0-9.]*' $OUT/*bench_line.json | paste - - - - -
head -8 $OUT/bench_kernel_stats.csv | cut -c1-150
