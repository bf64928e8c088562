mkdir -p gpurun_out/r4e
export PYTHONUNBUFFERED=1
R=$PWD
L=gpurun_out/r4e/tapwin_sched_ab.log
( timeout 300 python -m pytest tests/test_gpu_dcn.py -q -m gpu -k "tap_window" 2>&1 | tail -3 ) > gpurun_out/r4e/tests.log 2>&1
for rep in 1 2; do
  python scripts/bench_dcn_fwd_ab.py pinned 16 >> $L 2>&1
  for v in tw_unpinned tw_stagebar tw_valufirst; do
    EDVR_AMD_LIB=$R/edvr_amd/lib/variants/libedvr_amd_$v.so python scripts/bench_dcn_fwd_ab.py $v 16 >> $L 2>&1
  done
done
BENCH_ONLY=0 bash scripts/prof_pmc_kernel.sh r4e/tapwin_pmc dcn_tapwin_fwd_kernel python $R/scripts/bench_dcn_fwd_ab.py tapwin 16 > gpurun_out/r4e/tapwin_pmc.log 2>&1
BENCH_ONLY=0 EDVR_AMD_LIB=$R/edvr_amd/lib/variants/libedvr_amd_tw_unpinned.so bash scripts/prof_pmc_kernel.sh r4e/tapwin_unpinned_pmc dcn_tapwin_fwd_kernel python $R/scripts/bench_dcn_fwd_ab.py tapwin 16 > gpurun_out/r4e/tapwin_unpinned_pmc.log 2>&1
BENCH_ONLY=0 bash scripts/prof_pmc_kernel.sh r4e/halo3_pmc dcn_fused_fwd_kernel python $R/scripts/bench_dcn_fwd_ab.py halo3 3 > gpurun_out/r4e/halo3_pmc.log 2>&1
cat gpurun_out/r4e/tests.log; grep -v amdgpu.ids $L; cat gpurun_out/r4e/tapwin_pmc/pmc.json gpurun_out/r4e/tapwin_unpinned_pmc/pmc.json gpurun_out/r4e/halo3_pmc/pmc.json
