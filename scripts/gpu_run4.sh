mkdir -p gpurun_out/r4d
export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_train.py -q -m gpu -k "trajectory" -s 2>&1 | tail -12 ) > gpurun_out/r4d/traj.log 2>&1
BENCH_ONLY=0 bash scripts/prof_pmc_kernel.sh r4d/tapwin_pmc dcn_tapwin_fwd_kernel python scripts/bench_dcn_fwd_ab.py tapwin 16 > gpurun_out/r4d/tapwin_pmc.log 2>&1
BENCH_ONLY=0 bash scripts/prof_pmc_kernel.sh r4d/halo3_pmc dcn_fused_fwd_kernel python scripts/bench_dcn_fwd_ab.py halo3 3 > gpurun_out/r4d/halo3_pmc.log 2>&1
tail -6 gpurun_out/r4d/traj.log; cat gpurun_out/r4d/tapwin_pmc/pmc.json; cat gpurun_out/r4d/halo3_pmc/pmc.json
