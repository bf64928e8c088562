mkdir -p gpurun_out/r4h
export PYTHONUNBUFFERED=1
R=$PWD
bash scripts/prof_bench.sh r4h/prof_infer > gpurun_out/r4h/prof_infer.log 2>&1
bash scripts/prof_bench.sh r4h/prof_train --mode train > gpurun_out/r4h/prof_train.log 2>&1
bash scripts/prof_traffic.sh r4h/traffic > gpurun_out/r4h/traffic.log 2>&1
bash scripts/prof_pmc_f4.sh > gpurun_out/r4h/f4pmc.log 2>&1
BENCH_ONLY=0 bash scripts/prof_pmc_kernel.sh r4h/tapwin_pmc dcn_tapwin_fwd_kernel python $R/scripts/bench_dcn_fwd_ab.py tapwin 16 > gpurun_out/r4h/tapwin_pmc.log 2>&1
BENCH_ONLY=L1 bash scripts/prof_pmc_kernel.sh r4h/bwd_fused_pmc dcn_bwd_fused_kernel python $R/scripts/bench_dcn_bwd_ab.py 0.3 > gpurun_out/r4h/bwd_fused_pmc.log 2>&1
rm -f gpurun_out/r4h/prof_*/bench_kernel_trace.csv
ls gpurun_out/r4h gpurun_out/r4h/* gpurun_out/f4pmc | head -60; cat gpurun_out/r4h/traffic/traffic.json | head -40
