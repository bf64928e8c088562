#!/bin/bash
# round 6, call 4: conv kernel tests on the tree's default (round-5 loop, plain-fp32 transforms, sticky non-finite y_amax)
mkdir -p gpurun_out/r6
export PYTHONUNBUFFERED=1
O=gpurun_out/r6
timeout 600 python -m pytest tests/test_gpu_conv_f4s.py tests/test_gpu_conv_f4.py tests/test_gpu_conv.py -q --tb=short 2>&1 | tail -40 > $O/c4_tests.log
timeout 120 python scripts/bench_f4s_time.py default 2>&1 | grep -v amdgpu.ids >> $O/c4_tests.log
cat $O/c4_tests.log
