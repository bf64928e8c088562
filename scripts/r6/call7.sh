#!/bin/bash
# round 6, call 7: motion-field test, default bench line (motion legs, plain-fp32 transform)
mkdir -p gpurun_out/r6
export PYTHONUNBUFFERED=1
O=gpurun_out/r6
timeout 600 python -m pytest tests/test_gpu_edvr.py -x -q -k "motion_like" --tb=short 2>&1 | tail -15 > $O/c7_motion_test.log
timeout 1200 python bench.py > $O/c7_bench_stdout.log 2> $O/c7_bench_stderr.log
cp bench_full.json $O/c7_bench_full.json 2>/dev/null
cat $O/c7_motion_test.log; tail -c 2800 $O/c7_bench_stdout.log; tail -5 $O/c7_bench_stderr.log
