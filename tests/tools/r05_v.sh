# round 5: refresh the PMC traffic entries of the per-block, RCAB and fp8 kernels on the round's final sources
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_v; rm -rf $O; mkdir -p $O
RUMPY_NO_CHAIN=1 bash tests/tools/pmc_step.sh edsr > $O/pmc_step_edsr_per_block.txt 2>&1; cp gpurun_out/pmc_traffic_edsr.json $O/pmc_traffic_edsr_per_block.json
bash tests/tools/pmc_step.sh rcan > $O/pmc_step_rcan.txt 2>&1; cp gpurun_out/pmc_traffic_rcan.json $O/
bash tests/tools/pmc_step.sh rcan fp8 > $O/pmc_step_rcan_fp8.txt 2>&1; cp gpurun_out/pmc_traffic_rcan_fp8.json $O/
bash tests/tools/pmc_step.sh edsr fp8 > $O/pmc_step_edsr_fp8.txt 2>&1; cp gpurun_out/pmc_traffic_edsr_fp8.json $O/
tail -2 $O/pmc_step_edsr_per_block.txt $O/pmc_step_rcan.txt $O/pmc_step_rcan_fp8.txt $O/pmc_step_edsr_fp8.txt
