#!/bin/bash
mkdir -p gpurun_out/r6
export PYTHONUNBUFFERED=1
O=gpurun_out/r6
: > $O/c12_wd_ablate.log
timeout 200 python scripts/r6/bench_wgrad_time.py default 2>&1 | grep -v amdgpu.ids >> $O/c12_wd_ablate.log
EDVR_WGRAD_DIRECT_SPLIT=0 timeout 200 python scripts/r6/bench_wgrad_time.py winograd-split 2>&1 | grep -v amdgpu.ids >> $O/c12_wd_ablate.log
for v in wd_nocommit wd_nomfma wd_noloads wd_onlymfma; do
  EDVR_AMD_LIB=edvr_amd/lib/variants/libedvr_amd_$v.so timeout 200 python scripts/r6/bench_wgrad_time.py $v 2>&1 | grep -v amdgpu.ids >> $O/c12_wd_ablate.log
done
cat $O/c12_wd_ablate.log
