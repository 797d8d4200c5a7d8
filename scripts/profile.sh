#!/bin/bash
# rocprofv3 capture of the headline benchmark on the GPU box; summaries go to gpurun_out/prof_<tag>/ and are then
# copied into profiles/ by scripts/prof_summary.py.  Counters are collected in their own runs (kernel-trace only).
# usage: bash scripts/profile.sh <tag> [extra bench args]
set -u
TAG=${1:-r01}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BENCH="python bench.py --batch 1 --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --no-live-traffic --eager $*"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $BENCH > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o bench -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o bench -- $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU -d $OUT/pmc_sq -o bench -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA -d $OUT/pmc_mfma -o bench -- $BENCH > $OUT/pmc_mfma.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TA_TA_BUSY TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ -d $OUT/pmc_ta -o bench -- $BENCH > $OUT/pmc_ta.log 2>&1
python scripts/prof_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -size +20M -delete
cat $OUT/summary.txt
