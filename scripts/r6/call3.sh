#!/bin/bash
# round 6, call 3: A/B of the round-5 kernel (base), the tree's default (round-5 loop + sticky non-finite y_amax) and the plain-fp32 transform variants
mkdir -p gpurun_out/r6
export PYTHONUNBUFFERED=1
O=gpurun_out/r6
: > $O/c3_variants.log
timeout 120 python scripts/bench_f4s_time.py default 2>&1 | grep -v amdgpu.ids >> $O/c3_variants.log
for v in base m0pl m0plv; do
  EDVR_AMD_LIB=edvr_amd/lib/variants/libedvr_amd_$v.so timeout 120 python scripts/bench_f4s_time.py $v 2>&1 | grep -v amdgpu.ids >> $O/c3_variants.log
done
timeout 120 python scripts/bench_f4s_time.py default-again 2>&1 | grep -v amdgpu.ids >> $O/c3_variants.log
EDVR_AMD_LIB=edvr_amd/lib/variants/libedvr_amd_base.so timeout 120 python scripts/bench_f4s_time.py base-again 2>&1 | grep -v amdgpu.ids >> $O/c3_variants.log
EDVR_AMD_LIB=edvr_amd/lib/variants/libedvr_amd_m0pl.so timeout 300 python -m pytest tests/test_gpu_conv_f4s.py -x -q 2>&1 | tail -3 >> $O/c3_variants.log
timeout 300 python -m pytest tests/test_gpu_conv_f4s.py -x -q 2>&1 | tail -3 >> $O/c3_variants.log
cat $O/c3_variants.log
