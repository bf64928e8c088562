mkdir -p gpurun_out/r4a
cd /root/repo
export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_dcn.py -x -q 2>&1 | tail -25 ) > gpurun_out/r4a/test_dcn.log 2>&1
( timeout 420 python scripts/bench_dcn_sigma_sweep.py --quick ) > gpurun_out/r4a/sweep_quick.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_train.py -q -x -k "full_depth or unfollowed or trajectory" -s 2>&1 | tail -30 ) > gpurun_out/r4a/test_train_new.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_edvr.py tests/test_gpu_conv_f4.py tests/test_gpu_glue.py tests/test_gpu_graphs.py tests/test_gpu_compat_ext.py -q 2>&1 | tail -25 ) > gpurun_out/r4a/test_touched.log 2>&1
( timeout 900 python bench.py ) > gpurun_out/r4a/bench_default.json 2> gpurun_out/r4a/bench_default.err
tail -3 gpurun_out/r4a/test_dcn.log; cat gpurun_out/r4a/sweep_quick.log; tail -5 gpurun_out/r4a/test_train_new.log; tail -3 gpurun_out/r4a/test_touched.log; head -c 600 gpurun_out/r4a/bench_default.json
