#!/bin/bash
# Issue / LDS / vector-memory counters of the split-operand F(4x4) kernel on the n = 50 trunk layer, separate --pmc passes, each under
# its own timeout (a pass with the TCC_* counters aborted inside rocprofv3 and hung for the rest of the call's limit):
#   scripts/prof_pmc_f4s.sh NAME [KERNEL_SUBSTRING] [f4s|f4]   ->  gpurun_out/NAME/pmc.json
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; KEY=${2:-winograd_f4s_kernel}; ALGO=${3:-f4s}
OUT=$R/gpurun_out/$NAME; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/scripts/bench_f4s_one.py 50 128 180 320 128 $ALGO"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_MFMA" \
           "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" \
           "FETCH_SIZE" "WRITE_SIZE" ; do
  i=$((i+1))
  timeout 100 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p -- $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed / timed out: $set"
done
python $R/scripts/pmc_report.py "$KEY" $OUT/pmc.json $(ls $OUT/p*/*counter_collection.csv) > $OUT/report.log 2>&1
cat $OUT/report.log
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5 $OUT/p6
