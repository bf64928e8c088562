# round 5 evidence on the final kernel sources: rocprofv3 kernel stats of the inference and training steps, HBM traffic (FETCH_SIZE / WRITE_SIZE
# in separate --pmc passes + calibration copies), issue / LDS / vector-memory counters of the split F(4x4) kernel.  Every rocprofv3 under a timeout.
export PYTHONUNBUFFERED=1
bash scripts/prof_bench.sh r5/bench_edvr_l_infer > /dev/null 2>&1
bash scripts/prof_bench.sh r5/bench_edvr_l_train --mode train > /dev/null 2>&1
bash scripts/prof_traffic.sh r5/traffic > /dev/null 2>&1
bash scripts/prof_pmc_f4s.sh r5/pmc_f4s_final > /dev/null 2>&1
ls gpurun_out/r5/bench_edvr_l_infer gpurun_out/r5/bench_edvr_l_train gpurun_out/r5/traffic gpurun_out/r5/pmc_f4s_final
cut -c1-300 gpurun_out/r5/bench_edvr_l_infer/bench.json; cut -c1-300 gpurun_out/r5/bench_edvr_l_train/bench.json
head -6 gpurun_out/r5/bench_edvr_l_infer/*kernel_stats.csv | cut -c1-160
