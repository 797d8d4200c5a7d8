#!/bin/bash
# rocprofv3 counter passes over the warp + cost micro-benchmark (scripts/wbench.py); usage: bash scripts/prof_warp.sh <tag> [wbench args]
set -u
TAG=${1:-w}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD="python scripts/wbench.py --reps 5 $*"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o w -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU -d $OUT/pmc_sq -o w -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY -d $OUT/pmc_lds -o w -- $CMD > $OUT/pmc_lds.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TA_TA_BUSY TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ -d $OUT/pmc_ta -o w -- $CMD > $OUT/pmc_ta.log 2>&1
python scripts/prof_summary.py $OUT 2>&1 | grep -v "^$" | grep "warp_cost\|====\|SQ_\|GRBM\|TA_\|TCP_\|kernel  " 
find $OUT -name "*.db" -delete
