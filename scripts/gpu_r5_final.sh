# round 5, final tree: HBM traffic (FETCH_SIZE / WRITE_SIZE in separate --pmc passes + calibration copies) FIRST, copied to where bench.py
# reads it (same kernel-source hash), then the default bench line, rocprofv3 kernel stats of the inference and training steps, counters
# of the split DCN forward and the split dW product.  Every rocprofv3 under a timeout.
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r5f
bash scripts/prof_traffic.sh r5f/traffic > /dev/null 2>&1
cp gpurun_out/r5f/traffic/traffic.json profiles/r5/traffic_edvr_l_x4_t5_180x320.json
( timeout 400 python bench.py 2> gpurun_out/r5f/bench_default.err | tail -1 ) > gpurun_out/r5f/bench_default_run.json
bash scripts/prof_bench.sh r5f/bench_edvr_l_infer > /dev/null 2>&1
bash scripts/prof_bench.sh r5f/bench_edvr_l_train --mode train > /dev/null 2>&1
bash scripts/prof_pmc_kernel.sh r5f/pmc_dcn_split dcn_tapwin_split_fwd_kernel python $PWD/scripts/bench_dcn_split.py > /dev/null 2>&1
bash scripts/prof_pmc_kernel.sh r5f/pmc_gemm_split gemm_nt_split_kernel python $PWD/scripts/bench_gemm_split.py > /dev/null 2>&1
cut -c1-260 gpurun_out/r5f/bench_default_run.json; echo
cut -c1-200 gpurun_out/r5f/bench_edvr_l_infer/bench.json; echo; cut -c1-200 gpurun_out/r5f/bench_edvr_l_train/bench.json; echo
