#!/bin/bash
# collect_evidence.sh SRC_DIR ROUND   e.g. gpurun_out/r4j r4 : copy what scripts/gpu_evidence.sh wrote into profiles/ROUND/
set -e
S=$1; D=profiles/$2
cp $S/bench_default.json $D/bench_default_run.json
cp $S/prof_infer/bench.json $D/bench_edvr_l_infer.json; cp $S/prof_infer/bench_kernel_stats.csv $D/bench_edvr_l_infer_kernel_stats.csv
cp $S/prof_train/bench.json $D/bench_edvr_l_train.json; cp $S/prof_train/bench_kernel_stats.csv $D/bench_edvr_l_train_kernel_stats.csv
cp $S/traffic/traffic.json $D/traffic_edvr_l_x4_t5_180x320.json
cp $S/winograd_f4_micro_pmc.json $D/winograd_f4_micro_pmc.json
cp $S/tapwin_pmc/pmc.json $D/dcn_tapwin_micro_pmc.json; cp $S/bwd_fused_pmc/pmc.json $D/dcn_bwd_fused_micro_pmc.json
grep -v amdgpu.ids $S/dcn_sigma_sweep.log > $D/dcn_sigma_sweep.log; grep -v amdgpu.ids $S/dcn_fwd_shapes.log > $D/dcn_fwd_shapes.log
