mkdir -p gpurun_out/r5
export PYTHONUNBUFFERED=1
TAG=${1:-v3}
( timeout 600 python -m pytest tests/test_gpu_conv_f4s.py -q -x 2>&1 | tail -30 ) > gpurun_out/r5/test_f4s_$TAG.log 2>&1
( timeout 300 python scripts/bench_f4s.py $TAG ) > gpurun_out/r5/bench_f4s_$TAG.log 2>&1
tail -30 gpurun_out/r5/test_f4s_$TAG.log; cat gpurun_out/r5/bench_f4s_$TAG.log
