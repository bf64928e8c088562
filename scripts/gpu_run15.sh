mkdir -p gpurun_out/r4o
R=$PWD
L=gpurun_out/r4o/tapwin_prebarrier_ab.log
( timeout 300 python -m pytest tests/test_gpu_dcn.py -q -m gpu 2>&1 | tail -3 ) > gpurun_out/r4o/tests.log 2>&1
( EDVR_AMD_LIB=$R/edvr_amd/lib/variants/libedvr_amd_tw_wlate.so timeout 300 python scripts/check_dcn_variant.py 2>&1 | tail -2 ) > gpurun_out/r4o/tests_wlate.log 2>&1
for rep in 1 2; do
  python scripts/bench_dcn_fwd_ab.py pre_barrier 16 >> $L 2>&1
  EDVR_AMD_LIB=$R/edvr_amd/lib/variants/libedvr_amd_tw_wlate.so python scripts/bench_dcn_fwd_ab.py weights_late 16 >> $L 2>&1
  EDVR_AMD_LIB=$R/edvr_amd/lib/variants/libedvr_amd_tw_head.so python scripts/bench_dcn_fwd_ab.py round_start 16 >> $L 2>&1
done
cat gpurun_out/r4o/tests.log gpurun_out/r4o/tests_wlate.log; grep -v amdgpu.ids $L
