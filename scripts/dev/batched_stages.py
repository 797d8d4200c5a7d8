#!/usr/bin/env python3
"""Each stage of the MVSNet hot path as ONE launch (chain) over a batch of B reference views against B = 1: does batching the
latency-bound stages (the eight small layers, tail sweep, softargmin) beat running the views' launches side by side on streams?
us per launch(-chain) and per view.  Usage: python scripts/dev/batched_stages.py [--dtype bf16] [--batches 1,2,3,4,6]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as Bn
from wild_deep_mvs_amd import _lib as L, ops

ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--batches", default="1,2,3,4,6")
args = ap.parse_args()
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dt = Bn.DTYPES[args.dtype]
rows = {}
for nb in [int(x) for x in args.batches.split(",")]:
    net, sd, feats, fcl, proj_d, dv_d, _, _ = Bn.build_inputs(dev, 0, dt, nb)
    net.batch_streams = False
    ly = net.cost_regularization.engine_layers(dt)
    cams = ops.proj_cams_device(proj_d.float().contiguous(), 0)
    with torch.no_grad():
        cost = ops.warp_cost(fcl[0], fcl[1:], cams, dv_d, geom=L.GEOM_PROJ, cost=L.COST_VARIANCE, out_dtype=dt)
        c0 = ops.conv3d(cost, ly["conv0"])
        def small():
            c2 = ops.conv3d(ops.conv3d(c0, ly["conv1"]), ly["conv2"])
            c4 = ops.conv3d(ops.conv3d(c2, ly["conv3"]), ly["conv4"])
            c6 = ops.conv3d(ops.conv3d(c4, ly["conv5"]), ly["conv6"])
            u7 = ops.conv3d(c6, ly["conv7"], skip=c4)
            return ops.conv3d(u7, ly["conv9"], skip=c2)
        u9 = small()
        logits = ops.tail_sweep(u9, ly["conv11"], ly["prob"], skip=c0)
        dvf = dv_d.float().contiguous()
        stages = {
            "warp": lambda: ops.warp_cost(fcl[0], fcl[1:], cams, dv_d, geom=L.GEOM_PROJ, cost=L.COST_VARIANCE, out=cost),
            "conv0": lambda: ops.conv3d(cost, ly["conv0"], out=c0),
            "small8": small,
            "tail": lambda: ops.tail_sweep(u9, ly["conv11"], ly["prob"], skip=c0),
            "softargmin": lambda: ops.softargmin(logits, dvf, want_conf=True, conf_mode=0),
            "whole": lambda: net.hot_path(fcl, proj_d, dv_d),
        }
        for k, fn in stages.items():
            fn(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                for _ in range(10):
                    fn()
            best = 1e9
            for _ in range(6):
                torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            rows.setdefault(k, {})[nb] = best / 10 * 1e6
    del net, fcl, cost, c0, u9, logits
    torch.cuda.empty_cache()
print("# us per launch (chain) of a batch of B views, and (per view); 10 launches back to back in one graph, best of 6")
for k, d in rows.items():
    print(f"{k:11s} " + "  ".join(f"B={nb}: {v:7.1f} ({v / nb:6.1f})" for nb, v in d.items()))
