# round 5, first GPU call: micro-benchmark of the f16 matrix pipe beside vector work, parity of the split-operand kernel, its speed
mkdir -p gpurun_out/r5
export PYTHONUNBUFFERED=1
( timeout 120 scripts/micro/mfma16_prices ) > gpurun_out/r5/micro_mfma16_prices.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_conv_f4s.py -q -x 2>&1 | tail -25 ) > gpurun_out/r5/test_f4s.log 2>&1
( timeout 600 python scripts/bench_f4s.py v1 ) > gpurun_out/r5/bench_f4s_v1.log 2>&1
cat gpurun_out/r5/micro_mfma16_prices.log; tail -25 gpurun_out/r5/test_f4s.log; cat gpurun_out/r5/bench_f4s_v1.log
