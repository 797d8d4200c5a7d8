#!/bin/bash
# PMC passes over scripts/kbench.py --only cvp (64 -> 64 conv2d / conv3d, both wave mappings); CSV outputs under gpurun_out/prof_cvpconv
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_cvpconv
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/scripts/kbench.py --only cvp --dtype f16 --reps 5"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU -d $OUT/sq -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA -d $OUT/mfma -- $CMD > $OUT/mfma.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE TA_TA_BUSY TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ -d $OUT/ta -- $CMD > $OUT/ta.log 2>&1
ls -R $OUT | head -30
