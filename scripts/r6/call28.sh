#!/bin/bash
mkdir -p gpurun_out/r6
export PYTHONUNBUFFERED=1
O=gpurun_out/r6
: > $O/c28_dcn_fwd_ablate.log
timeout 200 python scripts/r6/bench_dcn_fwd_time.py default 2>&1 | grep -v amdgpu.ids >> $O/c28_dcn_fwd_ablate.log
for v in tw_nogather tw_nosplit tw_nomfma tw_nodma; do
  EDVR_AMD_LIB=edvr_amd/lib/variants/libedvr_amd_$v.so timeout 200 python scripts/r6/bench_dcn_fwd_time.py $v 2>&1 | grep -v amdgpu.ids >> $O/c28_dcn_fwd_ablate.log
done
cat $O/c28_dcn_fwd_ablate.log
