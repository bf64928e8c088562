#!/bin/bash
# round 6, call 11: the direct split weight gradient: tests, A/B against the Winograd-domain split kernel
mkdir -p gpurun_out/r6
export PYTHONUNBUFFERED=1
O=gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_wgrad.py -x -q --tb=short 2>&1 | tail -15 > $O/c11_wgrad_tests.log
echo "# direct split (default)" > $O/c11_bench_wgrad.log
timeout 600 python scripts/bench_wgrad_split.py 2>&1 | grep -v amdgpu.ids >> $O/c11_bench_wgrad.log
echo "# EDVR_WGRAD_DIRECT_SPLIT=0: Winograd-domain split" >> $O/c11_bench_wgrad.log
EDVR_WGRAD_DIRECT_SPLIT=0 timeout 600 python scripts/bench_wgrad_split.py 2>&1 | grep -v amdgpu.ids >> $O/c11_bench_wgrad.log
cat $O/c11_wgrad_tests.log $O/c11_bench_wgrad.log
