mkdir -p gpurun_out/r4p
R=$PWD
L=gpurun_out/r4p/wgrad_b128_ab.log
( timeout 600 python -m pytest tests/test_gpu_wgrad.py -q -m gpu 2>&1 | tail -3 ) > gpurun_out/r4p/tests.log 2>&1
for rep in 1 2; do
  echo "b128 layout" >> $L; python scripts/bench_wgrad.py 0,1,2,4,5 >> $L 2>&1
  echo "previous" >> $L; EDVR_AMD_LIB=$R/edvr_amd/lib/variants/libedvr_amd_ww_head.so python scripts/bench_wgrad.py 0,1,2,4,5 >> $L 2>&1
done
cat gpurun_out/r4p/tests.log; grep -v amdgpu.ids $L
