#!/bin/bash
# Memory-hierarchy and issue counters of the split-operand F(4x4) kernel on the n = 50 trunk layer, separate --pmc passes:
#   scripts/prof_pmc_f4s.sh NAME   ->  gpurun_out/NAME/pmc.json
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; KEY=${2:-winograd_f4s_kernel}; ALGO=${3:-f4s}
OUT=$R/gpurun_out/$NAME; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/scripts/bench_f4s_one.py 50 128 180 320 128 $ALGO"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum" \
           "FETCH_SIZE WRITE_SIZE TCC_TAG_STALL_sum TCC_BUSY_avr" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p -- $CMD > $OUT/p$i.log 2>&1
done
python $R/scripts/pmc_report.py "$KEY" $OUT/pmc.json $(ls $OUT/p*/*counter_collection.csv) > $OUT/report.log 2>&1
cat $OUT/report.log
for f in $OUT/p*.log; do tail -2 $f; done
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5
