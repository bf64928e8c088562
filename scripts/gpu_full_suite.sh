# GPU box: the whole -m gpu suite + __graft_entry__.smoke().   gpurun --timeout 1500 -- "bash scripts/gpu_full_suite.sh"
mkdir -p gpurun_out/r5k
export PYTHONUNBUFFERED=1
( timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -15 ) > gpurun_out/r5k/test_gpu_all.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r5k/smoke.log 2>&1
tail -4 gpurun_out/r5k/test_gpu_all.log; tail -2 gpurun_out/r5k/smoke.log
