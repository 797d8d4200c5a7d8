#!/usr/bin/env python3
"""Dev: per-kernel averages of the counters of a `rocprofv3 --pmc ... -d DIR` run (rocpd database).  python scripts/dev/pmc_table.py DIR"""
import glob
import sqlite3
import sys

db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
q = ("select kernel_name, counter_name, avg(v), count(*) from (select kernel_name, dispatch_id, counter_name, sum(value) as v "
     "from counters_collection group by kernel_name, dispatch_id, counter_name) group by kernel_name, counter_name")
tab = {}
for k, c, v, n in con.execute(q):
    tab.setdefault(k, {})[c] = v
    tab[k]["n"] = n
names = sorted({c for t in tab.values() for c in t if c != "n"})
print("kernel".ljust(60), "n".rjust(4), *(c.replace("SQ_", "")[:14].rjust(15) for c in names))
for k, t in sorted(tab.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", kv[1].get("SQ_WAVE_CYCLES", 0))):
    print(k[:60].ljust(60), str(t["n"]).rjust(4), *(f"{t.get(c, 0):15.0f}" for c in names))
