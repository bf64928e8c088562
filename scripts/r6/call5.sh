#!/bin/bash
# round 6, call 5: scalar-register cycle counters of the split F(4x4) kernel (who waits for whom)
mkdir -p gpurun_out/r6
export PYTHONUNBUFFERED=1
O=gpurun_out/r6
for v in prof prof2; do
  EDVR_AMD_LIB=edvr_amd/lib/variants/libedvr_amd_$v.so timeout 120 python scripts/f4s_prof.py 2>&1 | grep -v amdgpu.ids > $O/c5_$v.log
done
timeout 120 python scripts/bench_f4s_time.py default 2>&1 | grep -v amdgpu.ids > $O/c5_time.log
cat $O/c5_prof.log $O/c5_prof2.log $O/c5_time.log
