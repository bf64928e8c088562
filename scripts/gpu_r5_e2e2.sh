mkdir -p gpurun_out/r5
export PYTHONUNBUFFERED=1
TAG=${1:-e2e2}
( timeout 1500 python -m pytest tests/test_gpu_conv_f4s.py tests/test_gpu_edvr.py tests/test_gpu_train.py tests/test_gpu_wgrad.py tests/test_gpu_optim.py -q -x 2>&1 | tail -25 ) > gpurun_out/r5/test_$TAG.log 2>&1
( timeout 600 python bench.py --mode train --train-steps 20 --no-cpu-baseline --no-stock-baseline ) > gpurun_out/r5/bench_${TAG}_train_f4s.log 2>&1
( EDVR_WINOGRAD_F4S=0 timeout 600 python bench.py --mode train --train-steps 20 --no-cpu-baseline --no-stock-baseline ) > gpurun_out/r5/bench_${TAG}_train_f4.log 2>&1
tail -25 gpurun_out/r5/test_$TAG.log; tail -2 gpurun_out/r5/bench_${TAG}_train_f4s.log | cut -c1-1800; tail -2 gpurun_out/r5/bench_${TAG}_train_f4.log | cut -c1-600
