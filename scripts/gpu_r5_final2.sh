# one call: the split DCN forward's tests + micro-benchmark on the changed slab order, then (only if the tests pass) the whole final evidence
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r5f
timeout 200 python -m pytest tests/test_gpu_dcn.py -q -x -k "split or tap_window" 2>&1 | tail -3 > gpurun_out/r5f/dcn_tests.log
cat gpurun_out/r5f/dcn_tests.log
grep -q " passed" gpurun_out/r5f/dcn_tests.log || exit 1
grep -q "failed" gpurun_out/r5f/dcn_tests.log && exit 1
timeout 100 python scripts/bench_dcn_split.py 2>&1 | tee gpurun_out/bench_dcn_split_v3.log
bash scripts/gpu_r5_final.sh
