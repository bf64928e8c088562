#!/bin/bash
# round 6, call 10: the tap-window forward's new fix-up pass: DCN tests, motion probe
mkdir -p gpurun_out/r6
export PYTHONUNBUFFERED=1
O=gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_dcn.py tests/test_gpu_compat_ext.py -x -q --tb=short 2>&1 | tail -8 > $O/c10_dcn_tests.log
timeout 600 python -m pytest tests/test_gpu_edvr.py -x -q --tb=short 2>&1 | tail -8 >> $O/c10_dcn_tests.log
timeout 600 python scripts/r6/motion_probe.py 2>&1 | grep -v amdgpu.ids > $O/c10_motion_probe.log
cat $O/c10_dcn_tests.log $O/c10_motion_probe.log
