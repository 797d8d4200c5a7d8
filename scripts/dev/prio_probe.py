#!/usr/bin/env python3
"""Do HIP stream priorities help the free-running step?  Each view = two hipGraphs: FRONT (cameras, warp + cost, conv0) and BACK (eight
small layers, tail sweep, softargmin), chained by events (front(n) -> back(n) -> front(n + 1): the view's own order is unchanged).
Arms: both on one stream per view (= graph.ViewPipeline's schedule), on two streams of equal priority, back on a HIGH-priority stream
(its short latency-bound kernels are dispatched ahead of the queued warp workgroups), front on the high-priority stream.
Usage: python scripts/dev/prio_probe.py [--batch 3]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as Bn
from wild_deep_mvs_amd import _lib as L, ops

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=3)
ap.add_argument("--steps", type=int, default=150)
args = ap.parse_args()
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dt = torch.bfloat16
NB = args.batch
net, sd, feats, fcl, proj_d, dv_d, _, _ = Bn.build_inputs(dev, 0, dt, NB)
ly = net.cost_regularization.engine_layers(dt)
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print(f"# stream priority range: least {lo}, greatest {hi}")
dvf = dv_d.float().contiguous()

def build(front_prio, back_prio, two_streams):
    views = []
    with torch.no_grad():
        for b in range(NB):
            s1 = torch.cuda.Stream(priority=front_prio)
            s2 = torch.cuda.Stream(priority=back_prio) if two_streams else s1
            cost = torch.empty((1, Bn.D, Bn.h, Bn.w, 32), dtype=dt, device=dev)
            c0 = torch.empty((1, Bn.D, Bn.h, Bn.w, 8), dtype=dt, device=dev)
            fb = [f[b:b + 1] for f in fcl]

            def front():
                cm = ops.proj_cams_device(proj_d[b:b + 1].float().contiguous(), 0)
                ops.warp_cost(fb[0], fb[1:], cm, dv_d[b:b + 1], geom=L.GEOM_PROJ, cost=L.COST_VARIANCE, out=cost)
                ops.conv3d(cost, ly["conv0"], out=c0)

            def back():
                c2 = ops.conv3d(ops.conv3d(c0, ly["conv1"]), ly["conv2"])
                c4 = ops.conv3d(ops.conv3d(c2, ly["conv3"]), ly["conv4"])
                c6 = ops.conv3d(ops.conv3d(c4, ly["conv5"]), ly["conv6"])
                u7 = ops.conv3d(c6, ly["conv7"], skip=c4)
                u9 = ops.conv3d(u7, ly["conv9"], skip=c2)
                logits = ops.tail_sweep(u9, ly["conv11"], ly["prob"], skip=c0)
                return ops.softargmin(logits, dvf[b:b + 1], want_conf=True, conf_mode=0)
            gs = []
            for fn, st in ((front, s1), (back, s2)):
                with torch.cuda.stream(st):
                    fn(); torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
                        out = fn()
                gs.append((g, out))
            views.append((s1, s2, gs[0][0], gs[1][0], gs[1][1], (cost, c0, fb)))       # (the graphs hold raw pointers: keep their buffers alive)
    return views

def run(views, n):
    for _ in range(n):
        for s1, s2, gf, gb, _, _ in views:
            with torch.cuda.stream(s1):
                if s2 is not s1:
                    s1.wait_stream(s2)          # front(n + 1) after back(n): c0 is single-buffered
                gf.replay()
            with torch.cuda.stream(s2):
                if s2 is not s1:
                    s2.wait_stream(s1)
                gb.replay()

arms = {"one stream per view": (0, 0, False), "two streams, equal priority": (0, 0, True),
        "back (small layers, tail, softargmin) on a high-priority stream": (lo, hi, True),
        "front (warp, conv0) on a high-priority stream": (hi, lo, True)}
built = {k: build(*v) for k, v in arms.items()}
ref = None
for k, views in built.items():
    run(views, 2); torch.cuda.synchronize()
    d = torch.cat([v[4]["depth"] for v in views], 0)
    if ref is None: ref = d.clone()
    assert torch.equal(d, ref), k
acc = {k: [] for k in arms}
for r in range(4):
    for k, views in built.items():
        run(views, 30); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(views, args.steps); torch.cuda.synchronize()
        if r: acc[k].append((time.perf_counter() - t0) / args.steps * 1e3)
for k, v in acc.items():
    print(f"{k:70s} {sorted(v)[len(v) // 2]:.4f} ms per {NB}-view step  [{', '.join(f'{x:.4f}' for x in v)}]")
