#!/bin/bash
# round 6, evidence on the current tree: the whole -m gpu suite + smoke(), HBM traffic (separate --pmc passes + calibration), the default
# bench line, rocprofv3 kernel stats of the inference and training steps, counters of the split F(4x4) kernel.  Every rocprofv3 under a timeout.
export PYTHONUNBUFFERED=1
O=gpurun_out/r6f
mkdir -p $O profiles/r6
( timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -15 ) > $O/test_gpu_all.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
tail -3 $O/test_gpu_all.log; tail -2 $O/smoke.log
bash scripts/prof_traffic.sh r6f/traffic > /dev/null 2>&1
cp $O/traffic/traffic.json profiles/r6/traffic_edvr_l_x4_t5_180x320.json
( timeout 900 python bench.py 2> $O/bench_default.err | tail -1 ) > $O/bench_default_run.json
cp bench_full.json $O/bench_full.json 2>/dev/null
bash scripts/prof_bench.sh r6f/bench_edvr_l_infer > /dev/null 2>&1
bash scripts/prof_bench.sh r6f/bench_edvr_l_train --mode train > /dev/null 2>&1
bash scripts/prof_pmc_f4s.sh r6f/pmc_f4s > /dev/null 2>&1
# (the wave-cycle counters of the split F(4x4) kernel: profiles/r6/f4s_wave_cycle_counters*.log, scripts/f4s_prof.py on a -DF4S_PROF build)
cut -c1-400 $O/bench_default_run.json; echo
cut -c1-200 $O/bench_edvr_l_infer/bench.json; echo; cut -c1-200 $O/bench_edvr_l_train/bench.json; echo
rm -rf $O/bench_edvr_l_infer/*trace*.csv $O/bench_edvr_l_train/*trace*.csv 2>/dev/null
ls -la $O $O/bench_edvr_l_infer | head -40
