export PYTHONUNBUFFERED=1
bash scripts/prof_bench.sh r5/bench_edvr_l_infer > /dev/null 2>&1
bash scripts/prof_bench.sh r5/bench_edvr_l_train --mode train > /dev/null 2>&1
bash scripts/prof_pmc_kernel.sh r5/pmc_wgrad_split winograd_wgrad_split_kernel python $PWD/scripts/bench_wgrad_split.py > /dev/null 2>&1
bash scripts/prof_pmc_kernel.sh r5/pmc_dcn_split dcn_tapwin_split_fwd_kernel python $PWD/scripts/bench_dcn_split.py > /dev/null 2>&1
cut -c1-200 gpurun_out/r5/bench_edvr_l_infer/bench.json; cut -c1-200 gpurun_out/r5/bench_edvr_l_train/bench.json
head -8 gpurun_out/r5/bench_edvr_l_infer/*kernel_stats.csv | cut -c1-150; head -8 gpurun_out/r5/bench_edvr_l_train/*kernel_stats.csv | cut -c1-150
cat gpurun_out/r5/pmc_wgrad_split/pmc.json | head -40; cat gpurun_out/r5/pmc_dcn_split/pmc.json | head -40
