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
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("trace",):
    for f in glob.glob(f"{out}/{sub}/**/*kernel_stats.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "warp_cost" in row["Name"]:
                print("stats", row["Name"][:60], row["Calls"], "avg_ns", row["AverageNs"])
for sub in ("pmc_sq", "pmc_lds", "pmc_ta"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "warp_cost" in row["Kernel_Name"]:
                acc[row["Kernel_Name"][:50]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in acc.items():
        for c, v in sorted(d.items()):
            print(sub, k, c, f"{sum(v)/len(v):.4g}", f"(n={len(v)})")
PY
find $OUT -name "*.db" -size +20M -delete
