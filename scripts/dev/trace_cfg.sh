#!/bin/bash
# usage: bash scripts/dev/trace_cfg.sh <cfg> [n] [eager]: every kernel of the forward, per forward (calls / n)
CFG=$1; N=${2:-5}
OUT=gpurun_out/trace_cfg$CFG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace -d $OUT -o t -- python scripts/dev/trace_cfg.py $CFG $N $3 > $OUT/log.txt 2>&1
python - <<PY
import sqlite3, glob
db = glob.glob("$OUT/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
ev = con.execute("select name, start, end from kernels order by start").fetchall()
segs, cur = [], [ev[0]]
for e in ev[1:]:
    if e[1] - cur[-1][2] > 150e6:
        segs.append(cur); cur = []
    cur.append(e)
segs.append(cur)
last = segs[-1]
span = (last[-1][2] - last[0][1]) / 1e6
agg = {}
for nm, s_, e_ in last:
    a = agg.setdefault(nm, [0, 0]); a[0] += 1; a[1] += e_ - s_
tot = sum(a[1] for a in agg.values())
print(f"{len(segs)} forwards in the trace; the last one: {len(last)} kernels, {tot / 1e6:.3f} ms of kernel time in a span of {span:.3f} ms")
for nm, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    nm = nm.replace("void pscv::", "").replace("pscv::", "")[:110]
    print(f"{nm:110s} x{a[0]:4d} avg {a[1] / a[0] / 1e3:8.1f} us  {a[1] / 1e3:8.1f} us {100 * a[1] / tot:5.1f}%")
PY
find $OUT -name "*.db" -delete
