#!/bin/bash
# PMC counters of ONE kernel of a micro-benchmark, in separate rocprofv3 --pmc passes (never combined with trace domains other
# than --kernel-trace):  scripts/prof_pmc_kernel.sh NAME KERNEL_SUBSTRING command...   ->  gpurun_out/NAME/pmc.json
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; KEY=$2; shift 2
OUT=$R/gpurun_out/$NAME; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p1 -o p -- "$@" > $OUT/p1.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -o p -- "$@" > $OUT/p2.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_IFETCH --output-format csv -d $OUT/p3 -o p -- "$@" > $OUT/p3.log 2>&1
python $R/scripts/pmc_report.py "$KEY" $OUT/pmc.json $(ls $OUT/p*/*counter_collection.csv) > $OUT/report.log 2>&1
tail -25 $OUT/report.log
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
