#!/bin/bash
mkdir -p gpurun_out/r6
export PYTHONUNBUFFERED=1
O=gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_wgrad.py -x -q --tb=short 2>&1 | tail -5 > $O/c13_wd.log
timeout 200 python scripts/r6/bench_wgrad_time.py default 2>&1 | grep -v amdgpu.ids >> $O/c13_wd.log
EDVR_WGRAD_DIRECT_SPLIT=0 timeout 200 python scripts/r6/bench_wgrad_time.py winograd-split 2>&1 | grep -v amdgpu.ids >> $O/c13_wd.log
timeout 200 python scripts/r6/bench_wgrad_time.py default-again 2>&1 | grep -v amdgpu.ids >> $O/c13_wd.log
cat $O/c13_wd.log
