mkdir -p gpurun_out/r5
export PYTHONUNBUFFERED=1
TAG=${1:-v2}
( timeout 900 python -m pytest tests/test_gpu_conv_f4s.py -q 2>&1 | tail -15 ) > gpurun_out/r5/test_f4s_$TAG.log 2>&1
( timeout 600 python scripts/bench_f4s.py $TAG ) > gpurun_out/r5/bench_f4s_$TAG.log 2>&1
( EDVR_AMD_LIB=edvr_amd/lib/variants/libedvr_amd_f4strace.so SLOTS=30 timeout 300 python scripts/f4s_trace.py ) > gpurun_out/r5/f4s_trace_$TAG.log 2>&1
tail -5 gpurun_out/r5/test_f4s_$TAG.log; cat gpurun_out/r5/bench_f4s_$TAG.log; cat gpurun_out/r5/f4s_trace_$TAG.log
