#!/bin/bash
# scripts/prof_pmc.sh NAME KERNEL_SUBSTRING <command...>: rocprofv3 kernel stats + two separate --pmc passes (kernel-trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; KEY=$2; shift 2
OUT=$R/gpurun_out/$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- "$@" > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc1 -o p -- "$@" > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -o p -- "$@" > $OUT/pmc2.log 2>&1
python $R/scripts/pmc_report.py "$KEY" $OUT/pmc.json $(ls $OUT/pmc1/*counter_collection.csv $OUT/pmc2/*counter_collection.csv)
cp $(ls $OUT/stats/*kernel_stats.csv) $OUT/kernel_stats.csv
rm -rf $OUT/pmc1 $OUT/pmc2 $OUT/stats
head -6 $OUT/kernel_stats.csv | cut -c1-200
