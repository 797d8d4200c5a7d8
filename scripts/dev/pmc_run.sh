#!/bin/bash
# usage: bash scripts/dev/pmc_run.sh <tag> "<counters>" <cmd...>: one rocprofv3 --pmc pass, per-kernel averages
TAG=$1; CNT=$2; shift 2
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --pmc $CNT -d $OUT -o p -- "$@" > $OUT/log.txt 2>&1
python - <<PY
import sqlite3, glob
for db in glob.glob("$OUT/**/*.db", recursive=True):
    con = sqlite3.connect(db)
    q = ("select kernel_name, counter_name, avg(v), count(*) from (select kernel_name, counter_name, dispatch_id, sum(value) as v "
         "from counters_collection group by kernel_name, counter_name, dispatch_id) group by kernel_name, counter_name")
    tab = {}
    for k, c, v, n in con.execute(q):
        tab.setdefault(k, {})[c] = (v, n)
    for k, d in tab.items():
        if "pscv" not in k: continue
        print(k.replace("void pscv::", "")[:100])
        for c, (v, n) in sorted(d.items()):
            print(f"    {c:34s} {v:16.1f}   (x{n})")
PY
find $OUT -name "*.db" -delete
