#!/bin/bash
# rocprofv3 kernel stats of scripts/bench_dcn_bwd_paths.py (GPU box, via gpurun): which kernels the two DCN-backward paths spend their time in
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r5f/dcn_bwd_paths; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python $R/scripts/bench_dcn_bwd_paths.py > $OUT/run.log 2>&1
grep "per backward call" $OUT/run.log
head -14 $OUT/*kernel_stats.csv | cut -c1-170
