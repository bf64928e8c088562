#!/bin/bash
# round 6, call 8: motion legs with the amplitude cap
mkdir -p gpurun_out/r6
export PYTHONUNBUFFERED=1
O=gpurun_out/r6
timeout 600 python -m pytest tests/test_gpu_edvr.py -x -q -k "motion_like" --tb=short 2>&1 | tail -15 > $O/c8_motion_test.log
timeout 1200 python bench.py --no-cpu-baseline --no-configs --no-target-4k --no-batch4 --no-fp32-leg > $O/c8_bench_stdout.log 2> $O/c8_bench_stderr.log
cp bench_full.json $O/c8_bench_full.json 2>/dev/null
cat $O/c8_motion_test.log; tail -c 2500 $O/c8_bench_stdout.log; tail -5 $O/c8_bench_stderr.log
