mkdir -p gpurun_out/r4g
export PYTHONUNBUFFERED=1
( timeout 600 python -m pytest tests/test_gpu_graphs.py tests/test_gpu_data.py -q -m gpu 2>&1 | tail -8 ) > gpurun_out/r4g/tests.log 2>&1
( timeout 900 python bench.py ) > gpurun_out/r4g/bench_default.json 2> gpurun_out/r4g/bench_default.err
bash scripts/prof_bench.sh r4g/prof_infer > gpurun_out/r4g/prof_infer.log 2>&1
bash scripts/prof_bench.sh r4g/prof_train --mode train > gpurun_out/r4g/prof_train.log 2>&1
tail -4 gpurun_out/r4g/tests.log; head -c 300 gpurun_out/r4g/bench_default.json; echo; tail -3 gpurun_out/r4g/bench_default.err; ls gpurun_out/r4g/prof_infer gpurun_out/r4g/prof_train
