#!/usr/bin/env python3
"""When do the parallel branches of a captured hipGraph START?  Toy without the engine: N branches, each a chain of K elementwise torch
kernels of ~25 us on its own tensor, forked from / joined into the capture stream (the pattern of MVSNet._hot_path_streams).
  run:   python scripts/dev/graph_branch_start.py run N [K]          (under rocprofv3 --kernel-trace)
  show:  python scripts/dev/graph_branch_start.py show <dir> N [K]   prints, for the last replay, each branch's first-kernel start and last-kernel
         end relative to the replay's first kernel (branches are told apart by their queue)."""
import glob, os, sqlite3, sys


def run(n, k):
    import torch
    torch.cuda.set_device(0)
    xs = [torch.zeros(1 << 24, device="cuda") + i for i in range(n)]
    side = [torch.cuda.Stream() for _ in range(n)]

    def chain(x):
        for _ in range(k):
            x = x * 1.0001 + 0.5
        return x
    for x in xs:
        chain(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        outs = []
        for b in range(n):
            side[b].wait_stream(main)
            with torch.cuda.stream(side[b]):
                outs.append(chain(xs[b]))
        for b in range(n):
            main.wait_stream(side[b])
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()


def show(path, n, k):
    db = sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))[-1]
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    qcol = next(c for c in ("queue_id", "queue", "stream_id") if c in cols)
    rows = list(con.execute(f"select start, end, {qcol} from kernels order by start"))
    last = rows[-n * k:]
    t0 = min(r[0] for r in last)
    per = {}
    for s, e, q in last:
        per.setdefault(q, []).append((s, e))
    print(f"# {n} branches x {k} kernels, last replay ({len(last)} launches, {(max(r[1] for r in last) - t0) / 1e3:.1f} us):")
    for q, v in sorted(per.items(), key=lambda kv: min(x[0] for x in kv[1])):
        print(f"  queue {q}: {len(v):3d} kernels, first start {(min(x[0] for x in v) - t0) / 1e3:8.1f} us, last end {(max(x[1] for x in v) - t0) / 1e3:8.1f} us, "
              f"mean kernel {sum(e - s for s, e in v) / len(v) / 1e3:6.1f} us")


if __name__ == "__main__":
    n = int(sys.argv[2 if sys.argv[1] == "run" else 3])
    k = int(sys.argv[3 if sys.argv[1] == "run" else 4]) if len(sys.argv) > (3 if sys.argv[1] == "run" else 4) else 13
    run(n, k) if sys.argv[1] == "run" else show(sys.argv[2], n, k)
