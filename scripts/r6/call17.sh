#!/bin/bash
mkdir -p gpurun_out/r6
export PYTHONUNBUFFERED=1
O=gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_dcn.py tests/test_gpu_dcn1.py -x -q --tb=short 2>&1 | tail -4 > $O/c17_dcn.log
timeout 600 python scripts/r6/motion_probe_fp32.py 2>&1 | grep -v amdgpu.ids >> $O/c17_dcn.log
cat $O/c17_dcn.log
