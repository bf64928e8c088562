#!/bin/bash
mkdir -p gpurun_out/r6
export PYTHONUNBUFFERED=1
O=gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_dcn.py tests/test_gpu_dcn1.py tests/test_gpu_compat_ext.py tests/test_gpu_train.py -x -q --tb=short 2>&1 | tail -4 > $O/c22.log
timeout 300 python scripts/bench_dcn_bwd_paths.py 2>&1 | grep -v amdgpu.ids >> $O/c22.log
timeout 600 python scripts/r6/train_motion_table.py 2>&1 | grep -v amdgpu.ids >> $O/c22.log
cat $O/c22.log
