mkdir -p gpurun_out/r4l
R=$PWD
L=gpurun_out/r4l/tapwin_ablations.log
for rep in 1 2; do
  BENCH_ONLY=0 python scripts/bench_dcn_fwd_ab.py product 16 >> $L 2>&1
  for v in nobar nostate nodma notaps bare bare_nobar; do
    BENCH_ONLY=0 EDVR_AMD_LIB=$R/edvr_amd/lib/variants/libedvr_amd_twx_$v.so timeout 120 python scripts/bench_dcn_fwd_ab.py $v 16 >> $L 2>&1
  done
done
grep -v amdgpu.ids $L
