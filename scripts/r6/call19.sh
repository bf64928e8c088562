#!/bin/bash
mkdir -p gpurun_out/r6
export PYTHONUNBUFFERED=1
O=gpurun_out/r6
: > $O/c19_dcn_bwd_ablate.log
timeout 200 python scripts/r6/bench_dcn_bwd_lds.py default 2>&1 | grep -v amdgpu.ids >> $O/c19_dcn_bwd_ablate.log
for v in bw_noscatter bw_noglobal bw_nogather bw_neither; do
  EDVR_AMD_LIB=edvr_amd/lib/variants/libedvr_amd_$v.so timeout 200 python scripts/r6/bench_dcn_bwd_lds.py $v 2>&1 | grep -v amdgpu.ids >> $O/c19_dcn_bwd_ablate.log
done
cat $O/c19_dcn_bwd_ablate.log
