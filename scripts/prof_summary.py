#!/usr/bin/env python3
"""Summarise the rocprofv3 outputs of scripts/profile.sh: per-kernel duration statistics from the kernel trace and
per-kernel average PMC counters from the counter passes (rocpd sqlite databases).  Prints plain text."""
import glob
import os
import sqlite3
import sys


def dbs(d):
    return sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))


def short(name):
    name = name.replace("void pscv::", "").replace("pscv::", "")
    return name[:78]


def kernel_stats(path):
    con = sqlite3.connect(path)
    rows = con.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) "
                       "from kernels group by name order by 6 desc").fetchall()
    tot = sum(r[5] for r in rows) or 1
    print(f"{'kernel':80s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'share':>6s}")
    for r in rows:
        if "pscv" not in r[0] and r[5] / tot < 0.01:
            continue
        print(f"{short(r[0]):80s} {r[1]:6d} {r[2] / 1e3:9.1f} {r[3] / 1e3:9.1f} {r[4] / 1e3:9.1f} {100 * r[5] / tot:5.1f}%")


def pmc(path):
    con = sqlite3.connect(path)
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    if not cols:
        print("  (no counters_collection view)")
        return
    namecol = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else None)
    cntcol = "counter_name" if "counter_name" in cols else None
    valcol = "value" if "value" in cols else ("counter_value" if "counter_value" in cols else None)
    if not (namecol and cntcol and valcol):
        print("  columns:", cols)
        return
    disp = "dispatch_id" if "dispatch_id" in cols else None
    # a counter is reported per dimension instance (XCD/SE/...): sum the instances of one dispatch, then average dispatches
    q = (f"select {namecol}, {cntcol}, avg(v) from (select {namecol}, {cntcol}, {disp or '1'} as did, sum({valcol}) as v "
         f"from counters_collection group by {namecol}, {cntcol}, did) group by {namecol}, {cntcol}")
    table = {}
    for k, c, v in con.execute(q):
        if "pscv" in k:
            table.setdefault(short(k), {})[c] = v
    for k, d in sorted(table.items()):
        print(f"  {k}")
        for c, v in sorted(d.items()):
            print(f"      {c:32s} {v:18.1f}")


def main():
    root = sys.argv[1]
    for sub in sorted(os.listdir(root)):
        p = os.path.join(root, sub)
        if not os.path.isdir(p):
            continue
        for db in dbs(p):
            print(f"==== {sub}: {os.path.relpath(db, root)}")
            try:
                if sub == "trace":
                    kernel_stats(db)
                else:
                    pmc(db)
            except Exception as e:  # pragma: no cover
                print("  error:", e)


if __name__ == "__main__":
    main()
