# GPU box: every file profiles/rN/ is built from (kernel stats, traffic, PMC passes, sweeps, the default bench line) -> gpurun_out/r4j;
# then on the build host: scripts/collect_evidence.sh gpurun_out/r4j r4.   gpurun --timeout 1500 -- "bash scripts/gpu_evidence.sh"
mkdir -p gpurun_out/r4j
export PYTHONUNBUFFERED=1
R=$PWD
bash scripts/prof_bench.sh r4j/prof_infer > gpurun_out/r4j/prof_infer.log 2>&1
bash scripts/prof_bench.sh r4j/prof_train --mode train > gpurun_out/r4j/prof_train.log 2>&1
bash scripts/prof_traffic.sh r4j/traffic > gpurun_out/r4j/traffic.log 2>&1
bash scripts/prof_pmc_f4.sh > gpurun_out/r4j/f4pmc.log 2>&1
cp gpurun_out/f4pmc/pmc.json gpurun_out/r4j/winograd_f4_micro_pmc.json
BENCH_ONLY=0 bash scripts/prof_pmc_kernel.sh r4j/tapwin_pmc dcn_tapwin_fwd_kernel python $R/scripts/bench_dcn_fwd_ab.py tapwin 16 > gpurun_out/r4j/tapwin_pmc.log 2>&1
BENCH_ONLY=L1 bash scripts/prof_pmc_kernel.sh r4j/bwd_fused_pmc dcn_bwd_fused_kernel python $R/scripts/bench_dcn_bwd_ab.py 0.3 > gpurun_out/r4j/bwd_fused_pmc.log 2>&1
rm -f gpurun_out/r4j/prof_*/bench_kernel_trace.csv
( timeout 600 python scripts/bench_dcn_sigma_sweep.py ) > gpurun_out/r4j/dcn_sigma_sweep.log 2>&1
( python scripts/bench_dcn_fwd_ab.py tapwin 16; python scripts/bench_dcn_fwd_ab.py halo3 3 ) > gpurun_out/r4j/dcn_fwd_shapes.log 2>&1
( timeout 900 python bench.py ) > gpurun_out/r4j/bench_default.json 2> gpurun_out/r4j/bench_default.err
head -c 200 gpurun_out/r4j/bench_default.json; head -4 gpurun_out/r4j/prof_infer/bench_kernel_stats.csv | cut -c1-150
