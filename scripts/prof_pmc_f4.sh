R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/f4pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
CMD="python $R/scripts/bench_conv1_f4.py"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p1 -o p -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -o p -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_WAVE32_VALU SQ_THREAD_CYCLES_VALU SQ_IFETCH --output-format csv -d $OUT/p3 -o p -- $CMD > $OUT/p3.log 2>&1
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum --output-format csv -d $OUT/p4 -o p -- $CMD > $OUT/p4.log 2>&1
python $R/scripts/pmc_report.py winograd_f4_kernel $OUT/pmc.json $(ls $OUT/p*/*counter_collection.csv)
tail -3 $OUT/p3.log $OUT/p4.log
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
