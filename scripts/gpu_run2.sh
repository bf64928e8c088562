mkdir -p gpurun_out/r4b
export PYTHONUNBUFFERED=1
( timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -40 ) > gpurun_out/r4b/test_gpu_all.log 2>&1
( timeout 600 python scripts/bench_dcn_sigma_sweep.py ) > gpurun_out/r4b/dcn_sigma_sweep.log 2>&1
tail -5 gpurun_out/r4b/test_gpu_all.log; cat gpurun_out/r4b/dcn_sigma_sweep.log
