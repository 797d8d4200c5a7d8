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
    """All launches of the run, and the LAST THIRD of every kernel's launches: bench.py's per-kernel figures (`kernels_us`,
    `roofline.avg_us`) are HIP events around the launches of its final eager pass -- the first launches of a process run at ramping
    clocks and cold caches and pull the all-launch average up by 5-8 %."""
    con = sqlite3.connect(path)
    per = {}
    for name, start, end in con.execute("select name, start, end from kernels order by start"):
        per.setdefault(name, []).append(end - start)
    tot = sum(sum(v) for v in per.values()) or 1
    print(f"{'kernel':80s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'share':>6s} {'last-third avg_us':>18s}")
    for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        if "pscv" not in name and sum(v) / tot < 0.01:
            continue
        tail = v[len(v) - max(1, len(v) // 3):]
        print(f"{short(name):80s} {len(v):6d} {sum(v) / len(v) / 1e3:9.1f} {min(v) / 1e3:9.1f} {max(v) / 1e3:9.1f} {100 * sum(v) / tot:5.1f}% "
              f"{sum(tail) / len(tail) / 1e3:18.1f}")


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


def traffic_json(root):
    """Per-launch HBM bytes of the single-instance kernels from the FETCH_SIZE / WRITE_SIZE passes, corrected as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes: both counters are in KiB; FETCH_SIZE counts 64 B per 128-B
    request on gfx950 for wide coalesced reads, so it is doubled (calibrated here: the warp kernel then reads exactly
    its 5 feature maps); WRITE_SIZE matched the written bytes exactly (the cost volume) and is used as is."""
    import json
    vals = {}
    for sub, cname in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        for db in dbs(os.path.join(root, sub)):
            con = sqlite3.connect(db)
            q = ("select kernel_name, avg(v) from (select kernel_name, dispatch_id, sum(value) as v from counters_collection "
                 f"where counter_name = '{cname}' group by kernel_name, dispatch_id) group by kernel_name")
            for k, v in con.execute(q):
                vals.setdefault(k, {})[cname] = v
    keymap = (("warp_cost_lds_kernel", "warp_cost[0]"), ("warp_cost_q2_kernel", "warp_cost[0]"), ("warp_cost_kernel", "warp_cost[0]"), ("conv3d_sweep8_kernel", "conv3d[32->8,k3]"),
              ("conv3d_c1_kernel", "conv3d[8->1,k4]"), ("conv3d_c1_sweep_kernel", "conv3d[8->1,k4]"), ("conv3d_sweep_s2_kernel", "conv3d[8->16,k1]"),
              ("conv3d_t2p8_kernel", "conv3d[16->8,k5]"), ("conv3d_tail_kernel", "tail_sweep"), ("softargmin_kernel", "softargmin"))
    out = {}
    for k, d in vals.items():
        for sub, name in keymap:
            if sub in k and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
                out[name] = {"fetch_kib": d["FETCH_SIZE"], "write_kib": d["WRITE_SIZE"],
                             "hbm_bytes": (2.0 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0}
    with open(os.path.join(root, "traffic.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


def main():
    root = sys.argv[1]
    try:
        traffic_json(root)
    except Exception as e:  # pragma: no cover
        print("traffic.json: error", e)
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
