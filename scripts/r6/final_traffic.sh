#!/bin/bash
# after a kernel-source edit: HBM traffic of the bench command on the tree's hash, then the default bench line
export PYTHONUNBUFFERED=1
O=gpurun_out/r6g
mkdir -p $O profiles/r6
bash scripts/prof_traffic.sh r6g/traffic > /dev/null 2>&1
cp $O/traffic/traffic.json profiles/r6/traffic_edvr_l_x4_t5_180x320.json
( timeout 900 python bench.py 2> $O/bench_default.err | tail -1 ) > $O/bench_default_run.json
cp bench_full.json $O/bench_full.json 2>/dev/null
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
cut -c1-500 $O/bench_default_run.json; echo; tail -1 $O/smoke.log
