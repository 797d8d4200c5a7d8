#!/bin/bash
# PMC passes over the training step (dev aid); summaries under gpurun_out/prof_train_pmc/
set -u
OUT=gpurun_out/prof_train_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD="python scripts/bench_train.py --steps 2 --warmup 1"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU -d $OUT/pmc_sq -o bench -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_ATOMIC_RETURN -d $OUT/pmc_lds -o bench -- $CMD > $OUT/pmc_lds.log 2>&1
python scripts/prof_summary.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -size +20M -delete
grep -A 12 "warp_bwd_tile" $OUT/summary.txt | head -40
