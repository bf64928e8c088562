# round 5, final tree: the default bench line, rocprofv3 kernel stats of the inference and training steps, HBM traffic (FETCH_SIZE /
# WRITE_SIZE in separate --pmc passes + calibration copies), counters of the split DCN forward and the split dW product.
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r5f
( timeout 400 python bench.py 2> gpurun_out/r5f/bench_default.err | tail -1 ) > gpurun_out/r5f/bench_default_run.json
bash scripts/prof_bench.sh r5f/bench_edvr_l_infer > /dev/null 2>&1
bash scripts/prof_bench.sh r5f/bench_edvr_l_train --mode train > /dev/null 2>&1
bash scripts/prof_traffic.sh r5f/traffic > /dev/null 2>&1
bash scripts/prof_pmc_kernel.sh r5f/pmc_dcn_split dcn_tapwin_split_fwd_kernel python $PWD/scripts/bench_dcn_split.py > /dev/null 2>&1
bash scripts/prof_pmc_kernel.sh r5f/pmc_gemm_split gemm_nt_split_kernel python $PWD/scripts/bench_gemm_split.py > /dev/null 2>&1
cut -c1-260 gpurun_out/r5f/bench_default_run.json; echo
cut -c1-200 gpurun_out/r5f/bench_edvr_l_infer/bench.json; echo; cut -c1-200 gpurun_out/r5f/bench_edvr_l_train/bench.json; echo
ls gpurun_out/r5f/traffic gpurun_out/r5f/pmc_dcn_split gpurun_out/r5f/pmc_gemm_split
