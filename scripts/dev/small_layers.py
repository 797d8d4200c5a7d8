#!/usr/bin/env python3
"""The eight small layers of the MVSNet U-Net (conv1 ... conv9^T) at the headline size, one by one: us per launch (20 back to back in
one hipGraph, best of 5) for every value of a tuning knob, outputs compared bit for bit with the default's.
Usage: python scripts/dev/small_layers.py [--knob conv_small_nt --values 0,1,2,4] [--dtype bf16] [--batch 1]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as Bn
from wild_deep_mvs_amd import _lib as L, ops

ap = argparse.ArgumentParser()
ap.add_argument("--knob", default="conv_small_nt")
ap.add_argument("--values", default="0,1,2,4")
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--batch", type=int, default=1)
args = ap.parse_args()
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dt = Bn.DTYPES[args.dtype]
net, sd, feats, fcl, proj_d, dv_d, _, _ = Bn.build_inputs(dev, 0, dt, args.batch)
ly = net.cost_regularization.engine_layers(dt)
cams = ops.proj_cams_device(proj_d.float().contiguous(), 0)
with torch.no_grad():
    cost = ops.warp_cost(fcl[0], fcl[1:], cams, dv_d, geom=L.GEOM_PROJ, cost=L.COST_VARIANCE, out_dtype=dt)
    c0 = ops.conv3d(cost, ly["conv0"])
    c1 = ops.conv3d(c0, ly["conv1"]); c2 = ops.conv3d(c1, ly["conv2"])
    c3 = ops.conv3d(c2, ly["conv3"]); c4 = ops.conv3d(c3, ly["conv4"])
    c5 = ops.conv3d(c4, ly["conv5"]); c6 = ops.conv3d(c5, ly["conv6"])
    u7 = ops.conv3d(c6, ly["conv7"], skip=c4); u9 = ops.conv3d(u7, ly["conv9"], skip=c2)
    del cost
    layers = [("conv1 8->16 s2", lambda: ops.conv3d(c0, ly["conv1"])), ("conv2 16->16", lambda: ops.conv3d(c1, ly["conv2"])),
              ("conv3 16->32 s2", lambda: ops.conv3d(c2, ly["conv3"])), ("conv4 32->32", lambda: ops.conv3d(c3, ly["conv4"])),
              ("conv5 32->64 s2", lambda: ops.conv3d(c4, ly["conv5"])), ("conv6 64->64", lambda: ops.conv3d(c5, ly["conv6"])),
              ("conv7T 64->32", lambda: ops.conv3d(c6, ly["conv7"], skip=c4)), ("conv9T 32->16", lambda: ops.conv3d(u7, ly["conv9"], skip=c2))]

    def chain():
        a = ops.conv3d(ops.conv3d(c0, ly["conv1"]), ly["conv2"])
        b = ops.conv3d(ops.conv3d(a, ly["conv3"]), ly["conv4"])
        c = ops.conv3d(ops.conv3d(b, ly["conv5"]), ly["conv6"])
        return ops.conv3d(ops.conv3d(c, ly["conv7"], skip=b), ly["conv9"], skip=a)
    layers.append(("chain of 8", chain))
    vals = [int(v) for v in args.values.split(",")]
    ref = {}
    table = {}
    for v in vals:
        L.set_tuning(args.knob, v)
        for name, fn in layers:
            out = fn(); torch.cuda.synchronize()
            if v == vals[0]:
                ref[name] = out.clone()
            same = bool(torch.equal(out, ref[name]))
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                for _ in range(20):
                    fn()
            best = 1e9
            for _ in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            table.setdefault(name, []).append((best / 20 * 1e6, same))
    L.set_tuning(args.knob, vals[0])
print(f"# {args.knob}: us per launch (bits equal to the first column's?)  B = {args.batch}, {args.dtype}")
print(f"{'layer':18s} " + " ".join(f"{('=' + str(v)):>14s}" for v in vals))
for name, row in table.items():
    print(f"{name:18s} " + " ".join(f"{t:8.1f} {'same' if s else 'DIFF':>5s}" for t, s in row))
