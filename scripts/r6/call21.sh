#!/bin/bash
mkdir -p gpurun_out/r6
export PYTHONUNBUFFERED=1
O=gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_dcn.py tests/test_gpu_dcn1.py tests/test_gpu_compat_ext.py -x -q --tb=short 2>&1 | tail -4 > $O/c21_dcn_bwd_cas.log
timeout 200 python scripts/r6/bench_dcn_bwd_lds.py cas-loop 2>&1 | grep -v amdgpu.ids >> $O/c21_dcn_bwd_cas.log
timeout 300 python scripts/bench_dcn_bwd_paths.py 2>&1 | grep -v amdgpu.ids >> $O/c21_dcn_bwd_cas.log
cat $O/c21_dcn_bwd_cas.log
