#!/bin/bash
# Round-6 artefacts for profiles/: bench lines (headline at the default and at the driver's flags, per-block A/B, RCAN, blind QRCAN - each also fp8 -, the
# reference's 64-pixel crops, whole-image evaluation, MoCo), rocprofv3 --kernel-trace --stats of the headline / RCAN / 64-pixel / evaluation commands, PMC
# traffic (-> pmc_traffic.json entries) and the SQ counter pass.      usage (GPU box): bash tests/tools/r06_final.sh <tag>      -> gpurun_out/final_<tag>/
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final_$TAG
rm -rf $OUT; mkdir -p $OUT
cd $R
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench_err.log
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/driver_flags_bench_line.json 2>> $OUT/bench_err.log
RUMPY_NO_CHAIN=1 python3 bench.py --no-cpu-baseline > $OUT/edsr_per_block_bench_line.json 2>> $OUT/bench_err.log
python3 bench.py --precision fp8 --no-cpu-baseline > $OUT/edsr_fp8_bench_line.json 2>> $OUT/bench_err.log
python3 bench.py --model rcan > $OUT/rcan_bench_line.json 2>> $OUT/bench_err.log
python3 bench.py --model rcan --precision fp8 --no-cpu-baseline > $OUT/rcan_fp8_bench_line.json 2>> $OUT/bench_err.log
python3 bench.py --model blindqrcan --no-cpu-baseline > $OUT/blindqrcan_bench_line.json 2>> $OUT/bench_err.log
python3 bench.py --model blindqrcan --precision fp8 --no-cpu-baseline > $OUT/blindqrcan_fp8_bench_line.json 2>> $OUT/bench_err.log
python3 bench.py --lr-size 64 --batch 16 --steps 200 --warmup 30 --no-cpu-baseline > $OUT/edsr64_bench_line.json 2>> $OUT/bench_err.log
python3 bench.py --model rcan --lr-size 64 --batch 8 --steps 40 --warmup 8 --no-cpu-baseline > $OUT/rcan64_bench_line.json 2>> $OUT/bench_err.log
python3 bench.py --mode eval --steps 60 --warmup 6 > $OUT/eval_edsr_bench_line.json 2>> $OUT/bench_err.log
python3 bench.py --mode eval --model rcan --steps 20 --warmup 3 > $OUT/eval_rcan_bench_line.json 2>> $OUT/bench_err.log
python3 bench.py --model moco --steps 300 --warmup 30 > $OUT/moco_bench_line.json 2>> $OUT/bench_err.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_edsr -o p -- python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-as-called --settled-probe-ms 0 > $OUT/prof_edsr.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_rcan -o p -- python3 $R/bench.py --model rcan --steps 40 --warmup 10 --no-cpu-baseline --no-as-called --settled-probe-ms 0 > $OUT/prof_rcan.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_rcan64 -o p -- python3 $R/bench.py --model rcan --lr-size 64 --batch 8 --steps 40 --warmup 8 --no-cpu-baseline --no-as-called --settled-probe-ms 0 > $OUT/prof_rcan64.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_eval_rcan -o p -- python3 $R/bench.py --mode eval --model rcan --steps 20 --warmup 3 --no-cpu-baseline > $OUT/prof_eval_rcan.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_rcan_fp8 -o p -- python3 $R/bench.py --model rcan --precision fp8 --steps 40 --warmup 10 --no-cpu-baseline --no-as-called --settled-probe-ms 0 > $OUT/prof_rcan_fp8.log 2>&1
cd $R
for m in edsr rcan rcan64 eval_rcan rcan_fp8; do cp $(find $OUT/prof_$m -name '*kernel_stats.csv' | head -1) $OUT/${m}_kernel_stats.csv; rm -rf $OUT/prof_$m; done
bash tests/tools/pmc_step.sh edsr > $OUT/pmc_step_edsr.txt 2>&1
bash tests/tools/pmc_step.sh rcan > $OUT/pmc_step_rcan.txt 2>&1
bash tests/tools/pmc_step_sq.sh edsr > $OUT/pmc_sq_edsr.txt 2>&1
bash tests/tools/pmc_step_sq.sh rcan > $OUT/pmc_sq_rcan.txt 2>&1
cp gpurun_out/pmc_traffic_edsr.json gpurun_out/pmc_traffic_rcan.json $OUT/ 2>/dev/null
# the other PMC entries on the round's final sources: per-block launches, fp8 block / RCAB kernels
RUMPY_NO_CHAIN=1 bash tests/tools/pmc_step.sh edsr > $OUT/pmc_step_edsr_per_block.txt 2>&1; cp gpurun_out/pmc_traffic_edsr.json $OUT/pmc_traffic_edsr_per_block.json
bash tests/tools/pmc_step.sh rcan fp8 > $OUT/pmc_step_rcan_fp8.txt 2>&1; cp gpurun_out/pmc_traffic_rcan_fp8.json $OUT/ 2>/dev/null
bash tests/tools/pmc_step.sh edsr fp8 > $OUT/pmc_step_edsr_fp8.txt 2>&1; cp gpurun_out/pmc_traffic_edsr_fp8.json $OUT/ 2>/dev/null
SQ_KERNELS='wgrad_dma_kernel conv4dt_kernel conv_up_kernel tail_fwd_kernel conv4d_kernel' bash tests/tools/pmc_step_sq.sh edsr > $OUT/pmc_sq_edsr_others.txt 2>&1
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob('$OUT/*bench_line.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        print(os.path.basename(f), 'NO LINE'); continue
    r = d.get('roofline') or {}
    print('%-34s %10.2f %-13s %8.3f ms  kernel %s us frac %s traffic %s settled %s as_called %s' % (os.path.basename(f), d['value'], d['unit'], d['ms_per_step'], r.get('avg_launch_us'), r.get('frac'), r.get('traffic'), (d.get('settled') or {}).get('value'), (d.get('as_called') or {}).get('value')))
PY
for m in edsr rcan; do echo "== $m"; python3 tests/tools/prof_summary.py $OUT/${m}_kernel_stats.csv $( [ $m = edsr ] && echo 125 || echo 55 ) | head -16; done
tail -4 $OUT/pmc_step_edsr.txt $OUT/pmc_step_rcan.txt
tail -14 $OUT/pmc_sq_edsr.txt
tail -14 $OUT/pmc_sq_rcan.txt
