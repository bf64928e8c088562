mkdir -p gpurun_out/r5
export PYTHONUNBUFFERED=1
TAG=${1:-e2e1}
( timeout 900 python -m pytest tests/test_gpu_conv_f4s.py tests/test_gpu_edvr.py tests/test_gpu_conv_f4.py -q -x 2>&1 | tail -15 ) > gpurun_out/r5/test_$TAG.log 2>&1
( timeout 400 python bench.py --no-stock-baseline --no-train-leg --no-batch4 --no-target-4k --no-trained-like --no-configs ) > gpurun_out/r5/bench_${TAG}_f4s.log 2>&1
( EDVR_WINOGRAD_F4S=0 timeout 400 python bench.py --no-cpu-baseline --no-stock-baseline --no-train-leg --no-batch4 --no-target-4k --no-trained-like --no-configs ) > gpurun_out/r5/bench_${TAG}_f4.log 2>&1
tail -15 gpurun_out/r5/test_$TAG.log; tail -3 gpurun_out/r5/bench_${TAG}_f4s.log | cut -c1-3000; tail -2 gpurun_out/r5/bench_${TAG}_f4.log | cut -c1-1500
