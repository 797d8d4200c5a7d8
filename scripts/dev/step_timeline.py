#!/usr/bin/env python3
"""Timeline of ONE replay of the headline step (B views as parallel branches of one hipGraph): which kernels really run beside
which.  Two modes:
  run:    python scripts/dev/step_timeline.py run [--batch 3] [--stagger] [--tune k=v] [--replays 40]   (under rocprofv3 --kernel-trace)
  show:   python scripts/dev/step_timeline.py show <dir with the rocpd .db> [--launches-per-step N]
`show` prints, for the LAST complete step of the trace, every kernel with its start / end relative to the step's first kernel, the
queue it ran on, and a per-kernel-family summary of busy time and of the time it shared the GPU with other families."""
import argparse
import glob
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def run(args):
    import torch
    import bench as Bn
    from wild_deep_mvs_amd import _lib as L
    for kv in args.tune:
        k, v = kv.split("=")
        L.set_tuning(k, int(v))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    net, sd, feats, fcl, proj_d, dv_d, _, _ = Bn.build_inputs(dev, 0, Bn.DTYPES[args.dtype], args.batch)
    net.batch_streams = True
    net.batch_stagger = args.stagger
    with torch.no_grad():
        for _ in range(3):
            net.hot_path(fcl, proj_d, dv_d)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            net.hot_path(fcl, proj_d, dv_d)
        for _ in range(args.replays):
            g.replay()
        torch.cuda.synchronize()


def short(name):
    name = name.replace("void pscv::", "").replace("pscv::", "")
    cut = name.find("(")
    name = name if cut < 0 else name[:cut]
    return name[:60]


def family(name):
    s = short(name)
    return s.split("<")[0]


def show(args):
    paths = sorted(glob.glob(os.path.join(args.path, "**", "*.db"), recursive=True))
    if not paths:
        raise SystemExit(f"no .db under {args.path}")
    con = sqlite3.connect(paths[-1])
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    qcol = next((c for c in ("queue_id", "queue", "stream_id", "stream") if c in cols), None)
    rows = list(con.execute(f"select name, start, end{', ' + qcol if qcol else ''} from kernels order by start"))
    rows = [r for r in rows if "pscv" in r[0]]
    n = args.launches_per_step
    if n <= 0:
        # a step = the launches between two consecutive occurrences of the step's first kernel on its first queue: count kernels of
        # the last third of the trace and divide by the replays there (the caller passes --launches-per-step when this guess is off)
        names = [r[0] for r in rows]
        first = names[-1]
        n = len(names) - 1 - max(i for i, x in enumerate(names[:-1]) if x == first) if names.count(first) > 1 else len(names)
    step = rows[-n:]
    t0 = min(r[1] for r in step)
    t1 = max(r[2] for r in step)
    print(f"# last step of {paths[-1]}: {n} launches, {(t1 - t0) / 1e3:.1f} us from the first start to the last end")
    print(f"{'start_us':>9s} {'end_us':>9s} {'dur_us':>8s} {'queue':>6s}  kernel")
    for r in sorted(step, key=lambda r: r[1]):
        print(f"{(r[1] - t0) / 1e3:9.1f} {(r[2] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:8.1f} {str(r[3]) if qcol else '-':>6s}  {short(r[0])}")
    # per family: busy time (union of its intervals) and the part of it during which another family was running too
    fams = {}
    for r in step:
        fams.setdefault(family(r[0]), []).append((r[1], r[2]))

    def union(iv):
        iv = sorted(iv)
        out = []
        for a, b in iv:
            if out and a <= out[-1][1]:
                out[-1][1] = max(out[-1][1], b)
            else:
                out.append([a, b])
        return out

    def length(iv):
        return sum(b - a for a, b in iv)

    def intersect(x, y):
        out, i, j = [], 0, 0
        while i < len(x) and j < len(y):
            a, b = max(x[i][0], y[j][0]), min(x[i][1], y[j][1])
            if a < b:
                out.append([a, b])
            if x[i][1] < y[j][1]:
                i += 1
            else:
                j += 1
        return out
    print(f"\n{'family':40s} {'launches':>8s} {'sum_us':>9s} {'busy_us':>9s} {'beside other families_us':>25s}")
    for f, iv in sorted(fams.items(), key=lambda kv: -length(union(kv[1]))):
        u = union(iv)
        others = union([x for g, w in fams.items() if g != f for x in w])
        print(f"{f:40s} {len(iv):8d} {sum(b - a for a, b in iv) / 1e3:9.1f} {length(u) / 1e3:9.1f} {length(intersect(u, others)) / 1e3:25.1f}")
    allu = union([x for w in fams.values() for x in w])
    print(f"GPU busy (any engine kernel): {length(allu) / 1e3:.1f} us of {(t1 - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    r = sub.add_parser("run")
    r.add_argument("--batch", type=int, default=3)
    r.add_argument("--stagger", action="store_true")
    r.add_argument("--dtype", default="bf16")
    r.add_argument("--replays", type=int, default=40)
    r.add_argument("--tune", action="append", default=[])
    s = sub.add_parser("show")
    s.add_argument("path")
    s.add_argument("--launches-per-step", type=int, default=0)
    a = ap.parse_args()
    run(a) if a.cmd == "run" else show(a)
