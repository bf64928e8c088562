#!/bin/bash
# HBM traffic of the bench command: separate rocprofv3 --pmc passes for FETCH_SIZE and WRITE_SIZE (never combined with
# trace domains other than --kernel-trace), plus the same two passes over known-byte-count copies for calibration.
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=${1:-prof_traffic}; shift
OUT=$R/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o b -- python $R/bench.py --no-cpu-baseline --no-stock-baseline --no-train-leg --no-batch4 --no-target-4k --no-trained-like --no-configs --no-fp32-leg --no-roofline --steps 1 --warmup 1 "$@" > $OUT/$c.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/calib_$c -o c -- python $R/scripts/traffic_calib.py > $OUT/calib_$c.log 2>&1
done
F=$(ls $OUT/FETCH_SIZE/*counter_collection.csv); W=$(ls $OUT/WRITE_SIZE/*counter_collection.csv)
CF=$(ls $OUT/calib_FETCH_SIZE/*counter_collection.csv); CW=$(ls $OUT/calib_WRITE_SIZE/*counter_collection.csv)
python $R/scripts/traffic_report.py $OUT/traffic.json $F $W $CF $CW
rm -rf $OUT/FETCH_SIZE $OUT/WRITE_SIZE $OUT/calib_FETCH_SIZE $OUT/calib_WRITE_SIZE   # raw csvs are large; keep the summary
tail -2 $OUT/*.log | cut -c1-300
