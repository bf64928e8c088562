mkdir -p gpurun_out/r4n
R=$PWD
L=gpurun_out/r4n/tapwin_overlap_ab.log
( timeout 300 python -m pytest tests/test_gpu_dcn.py -q -m gpu 2>&1 | tail -3 ) > gpurun_out/r4n/tests.log 2>&1
for rep in 1 2; do
  python scripts/bench_dcn_fwd_ab.py state_overlap 16 >> $L 2>&1
  EDVR_AMD_LIB=$R/edvr_amd/lib/variants/libedvr_amd_tw_head.so python scripts/bench_dcn_fwd_ab.py previous 16 >> $L 2>&1
done
cat gpurun_out/r4n/tests.log; grep -v amdgpu.ids $L
