mkdir -p gpurun_out/r4f
export PYTHONUNBUFFERED=1
( timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -25 ) > gpurun_out/r4f/test_gpu_all.log 2>&1
( timeout 600 python scripts/bench_dcn_sigma_sweep.py ) > gpurun_out/r4f/dcn_sigma_sweep.log 2>&1
( timeout 900 python bench.py ) > gpurun_out/r4f/bench_default.json 2> gpurun_out/r4f/bench_default.err
tail -6 gpurun_out/r4f/test_gpu_all.log; grep -v amdgpu gpurun_out/r4f/dcn_sigma_sweep.log | cut -c1-330; head -c 400 gpurun_out/r4f/bench_default.json
