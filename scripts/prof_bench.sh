#!/bin/bash
# rocprofv3 kernel stats of the bench command (run on the GPU box via gpurun): scripts/prof_bench.sh NAME [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=${1:-prof_bench}; shift
OUT=$R/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python $R/bench.py --no-cpu-baseline --no-stock-baseline --no-train-leg --no-batch4 --no-target-4k --no-trained-like --no-configs --no-fp32-leg "$@" > $OUT/bench.log 2>&1
grep '^{"metric' $OUT/bench.log > $OUT/bench.json; cut -c1-400 $OUT/bench.json
ls $OUT
head -12 $OUT/*kernel_stats.csv | cut -c1-200
