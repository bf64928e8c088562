mkdir -p gpurun_out/r5
export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_conv_f4s.py -q 2>&1 | tail -40 ) > gpurun_out/r5/test_f4s.log 2>&1
( timeout 600 python scripts/bench_f4s.py v1 ) > gpurun_out/r5/bench_f4s_v1.log 2>&1
tail -40 gpurun_out/r5/test_f4s.log; cat gpurun_out/r5/bench_f4s_v1.log
