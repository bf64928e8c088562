#!/bin/bash
mkdir -p gpurun_out/r6
export PYTHONUNBUFFERED=1
O=gpurun_out/r6
: > $O/c25_wgrad_ablate.log
timeout 200 python scripts/r6/bench_wgrad_time.py default 2>&1 | grep -v amdgpu.ids >> $O/c25_wgrad_ablate.log
for v in wg_nostage wg_nomfma wg_nosplit wg_noldsw; do
  EDVR_AMD_LIB=edvr_amd/lib/variants/libedvr_amd_$v.so timeout 200 python scripts/r6/bench_wgrad_time.py $v 2>&1 | grep -v amdgpu.ids >> $O/c25_wgrad_ablate.log
done
cat $O/c25_wgrad_ablate.log
