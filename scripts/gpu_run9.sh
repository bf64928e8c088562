mkdir -p gpurun_out/r4i
export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_conv_f4.py tests/test_gpu_dcn.py tests/test_gpu_edvr.py -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r4i/tests.log 2>&1
( timeout 900 python bench.py ) > gpurun_out/r4i/bench_default.json 2> gpurun_out/r4i/bench_default.err
tail -3 gpurun_out/r4i/tests.log; head -c 300 gpurun_out/r4i/bench_default.json
