#!/bin/bash
# round 6, call 1: multiplying-wave split of the F(4x4) kernel vs the round-5 kernel (A/B, traces), and the compact bench line
mkdir -p gpurun_out/r6
export PYTHONUNBUFFERED=1
O=gpurun_out/r6
timeout 600 python -m pytest tests/test_gpu_conv_f4s.py -x -q 2>&1 | tail -5 > $O/c1_test_f4s.log
for v in default base; do
  if [ $v = default ]; then unset EDVR_AMD_LIB; else export EDVR_AMD_LIB=edvr_amd/lib/variants/libedvr_amd_$v.so; fi
  timeout 300 python scripts/bench_f4s.py $v 2>&1 | grep -v amdgpu.ids >> $O/c1_bench_f4s.log
done
for v in basetrace mwtrace; do
  EDVR_AMD_LIB=edvr_amd/lib/variants/libedvr_amd_$v.so timeout 120 python scripts/f4s_trace.py 2>&1 | grep -v amdgpu.ids > $O/c1_trace_$v.log
done
unset EDVR_AMD_LIB
timeout 900 python bench.py > $O/c1_bench_stdout.log 2> $O/c1_bench_stderr.log
cp bench_full.json $O/c1_bench_full.json 2>/dev/null
tail -c 3000 $O/c1_bench_stdout.log
cat $O/c1_test_f4s.log $O/c1_bench_f4s.log
