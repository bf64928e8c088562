#!/bin/bash
# round 6, call 2: multiplying-loop variants of the split F(4x4) kernel (A/B timing + barrier / position traces)
mkdir -p gpurun_out/r6
export PYTHONUNBUFFERED=1
O=gpurun_out/r6
: > $O/c2_variants.log
timeout 120 python scripts/bench_f4s_time.py default 2>&1 | grep -v amdgpu.ids >> $O/c2_variants.log
for v in m0 m1 m2p0 m2p1 m2pl m1pl m0pl; do
  EDVR_AMD_LIB=edvr_amd/lib/variants/libedvr_amd_$v.so timeout 120 python scripts/bench_f4s_time.py $v 2>&1 | grep -v amdgpu.ids >> $O/c2_variants.log
done
timeout 120 python scripts/bench_f4s_time.py default-again 2>&1 | grep -v amdgpu.ids >> $O/c2_variants.log
for v in m2trace m0trace m1trace; do
  EDVR_AMD_LIB=edvr_amd/lib/variants/libedvr_amd_$v.so timeout 120 python scripts/f4s_trace.py 2>&1 | grep -v amdgpu.ids > $O/c2_trace_$v.log
done
timeout 300 python -m pytest tests/test_gpu_conv_f4s.py -x -q 2>&1 | tail -3 >> $O/c2_variants.log
timeout 600 python -m pytest tests/test_gpu_edvr.py -x -q -k "inference_mode or heavy_tailed or batch_composition or overflow_guard or void_stale or magnitude_bounds or offset_check" 2>&1 | tail -15 > $O/c2_guard_tests.log
timeout 300 python -m pytest tests/test_gpu_graphs.py -x -q 2>&1 | tail -5 >> $O/c2_guard_tests.log
cat $O/c2_variants.log $O/c2_guard_tests.log
