#!/usr/bin/env python3
"""Feasibility probe of a two-stage pipeline of the MVSNet step: stream A runs [warp + cost, conv0] view after view (B = 1 launches),
stream B runs the rest of the U-Net + regression BATCHED over the B views of the previous step (one launch per layer for B views:
the small layers cost 78 us per view that way against 102).  No data dependency between the streams here (independent buffers): this
measures what the GPU makes of the mix, i.e. the ceiling of such a pipeline.  Per arm: ms per B views.
Usage: python scripts/dev/stage_pipeline_probe.py [--batch 3] [--dtype bf16] [--tune k=v]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as Bn
from wild_deep_mvs_amd import _lib as L, ops

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=3)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--tune", action="append", default=[])
args = ap.parse_args()
for kv in args.tune:
    k, v = kv.split("="); L.set_tuning(k, int(v))
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dt = Bn.DTYPES[args.dtype]
NB = args.batch
net, sd, feats, fcl, proj_d, dv_d, _, _ = Bn.build_inputs(dev, 0, dt, NB)
ly = net.cost_regularization.engine_layers(dt)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
with torch.no_grad():
    dvf = dv_d.float().contiguous()
    cams = ops.proj_cams_device(proj_d.float().contiguous(), 0)
    cost1 = torch.empty((1, Bn.D, Bn.h, Bn.w, 32), dtype=dt, device=dev)
    c0 = torch.empty((NB, Bn.D, Bn.h, Bn.w, 8), dtype=dt, device=dev)

    def front():                                   # stream A: the B views one after the other
        for b in range(NB):
            cm = ops.proj_cams_device(proj_d[b:b + 1].float().contiguous(), 0)
            ops.warp_cost(fcl[0][b:b + 1], [f[b:b + 1] for f in fcl[1:]], cm, dv_d[b:b + 1], geom=L.GEOM_PROJ, cost=L.COST_VARIANCE, out=cost1)
            ops.conv3d(cost1, ly["conv0"], out=c0[b:b + 1])

    c0b = torch.randn((NB, Bn.D, Bn.h, Bn.w, 8), device=dev).to(dt)

    def back():                                    # stream B: the rest, batched over the B views (of the previous step)
        c2 = ops.conv3d(ops.conv3d(c0b, ly["conv1"]), ly["conv2"])
        c4 = ops.conv3d(ops.conv3d(c2, ly["conv3"]), ly["conv4"])
        c6 = ops.conv3d(ops.conv3d(c4, ly["conv5"]), ly["conv6"])
        u7 = ops.conv3d(c6, ly["conv7"], skip=c4)
        u9 = ops.conv3d(u7, ly["conv9"], skip=c2)
        logits = ops.tail_sweep(u9, ly["conv11"], ly["prob"], skip=c0b)
        return ops.softargmin(logits, dvf, want_conf=True, conf_mode=0)

    def graph_of(fn, stream):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(stream):
            fn(); torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
                fn()
        return g
    gf, gb = graph_of(front, sa), graph_of(back, sb)

    def timeit(run, n=args.steps):
        run(10); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); run(n); torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / n)
        return best * 1e3

    def only_a(n):
        with torch.cuda.stream(sa):
            for _ in range(n): gf.replay()
    def only_b(n):
        with torch.cuda.stream(sb):
            for _ in range(n): gb.replay()
    def both(n):
        for _ in range(n):
            with torch.cuda.stream(sa): gf.replay()
            with torch.cuda.stream(sb): gb.replay()
    ta, tb, tab = timeit(only_a), timeit(only_b), timeit(both)
    print(f"B = {NB}, {args.dtype}: front (cams + warp + conv0, view after view) alone {ta:.4f} ms; back (8 small layers + tail sweep + softargmin, batched) alone {tb:.4f} ms; "
          f"both streams {tab:.4f} ms per {NB} views = {tab / NB * 1e3:.1f} us per view  (sum {ta + tb:.4f}, gain {(ta + tb) / tab:.3f})")
