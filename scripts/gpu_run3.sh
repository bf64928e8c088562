mkdir -p gpurun_out/r4c
export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_dcn.py tests/test_gpu_dcn1.py tests/test_gpu_compat_ext.py tests/test_gpu_wgrad.py tests/test_metrics.py tests/test_gpu_fullsize_properties.py -q -m gpu -k "trajectory or dcn or compat or wgrad or metrics or psnr or ssim or fullsize" -s 2>&1 | tail -30 ) > gpurun_out/r4c/tests.log 2>&1
for s in 0.3 0.5; do
  ( python scripts/bench_dcn_bwd_ab.py $s; EDVR_AMD_LIB=$PWD/edvr_amd/lib/variants/libedvr_amd_oldbwd.so python scripts/bench_dcn_bwd_ab.py $s ) >> gpurun_out/r4c/dcn_bwd_asm_dma_ab.log 2>&1
done
( python scripts/bench_dcn_fwd_ab.py tapwin 16; python scripts/bench_dcn_fwd_ab.py halo3 3 ) > gpurun_out/r4c/dcn_fwd_shapes.log 2>&1
tail -8 gpurun_out/r4c/tests.log; cat gpurun_out/r4c/dcn_bwd_asm_dma_ab.log gpurun_out/r4c/dcn_fwd_shapes.log
