#!/bin/bash
# rocprofv3 on the conv micro-benchmark: kernel stats + two PMC passes (run on the GPU box via gpurun).
set -x
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-prof_conv}
SH=${2:-0}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python $R/scripts/bench_conv.py $SH > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc1 -o pmc1 -- python $R/scripts/bench_conv.py $SH > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -o pmc2 -- python $R/scripts/bench_conv.py $SH > $OUT/pmc2.log 2>&1
ls -R $OUT | head -50
tail -n 3 $OUT/stats.log
