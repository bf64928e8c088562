mkdir -p gpurun_out/r4r
R=$PWD
L=gpurun_out/r4r/glue_ab.log
( timeout 600 python -m pytest tests/test_gpu_glue.py tests/test_gpu_wgrad.py tests/test_gpu_conv.py tests/test_gpu_edvr.py -q -m gpu -x 2>&1 | tail -5 ) > gpurun_out/r4r/tests.log 2>&1
for rep in 1 2; do
  python scripts/bench_glue.py new >> $L 2>&1
  EDVR_AMD_LIB=$R/edvr_amd/lib/variants/libedvr_amd_head_glue.so python scripts/bench_glue.py previous >> $L 2>&1
done
cat gpurun_out/r4r/tests.log; grep -v amdgpu.ids $L
