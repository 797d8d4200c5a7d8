#!/usr/bin/env python3
"""ms per 3-view step of the free-running per-view graphs (graph.ViewPipeline) for values of tuning knobs, one process, arms interleaved
over several rounds (medians).  Usage: python scripts/dev/knob_sweep_views.py "warp_ppd=32" "warp_ppd=48" "warp_ppd=48,sweep_dc=64" ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as Bn
from wild_deep_mvs_amd import _lib as L
from wild_deep_mvs_amd.graph import ViewPipeline

arms = [a for a in sys.argv[1:] if not a.startswith("--")] or ["default"]
rounds, steps = 3, 150
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
net, sd, feats, fcl, proj_d, dv_d, _, _ = Bn.build_inputs(dev, 0, torch.bfloat16, 3)
knobs = sorted({kv.split("=")[0] for a in arms if a != "default" for kv in a.split(",")})
base = {k: L.get_tuning(k) for k in knobs}
pipes, ref = {}, None
with torch.no_grad():
    for a in arms:
        for k in knobs:
            L.set_tuning(k, base[k])
        if a != "default":
            for kv in a.split(","):
                k, v = kv.split("="); L.set_tuning(k, int(v))
        pipes[a] = ViewPipeline(net, fcl, proj_d, dv_d)
        pipes[a].step(); d = pipes[a].results()[0]; torch.cuda.synchronize()
        if ref is None:
            ref = d.clone()
        if not torch.equal(d, ref):     # (knobs that change a summation order move the depth by fp32 rounding)
            err = float((d - ref).abs().max() / ref.abs().max())
            print(f"# {a}: depth differs from the first arm by {err:.2e} (max norm)")
            pass
    for k in knobs:
        L.set_tuning(k, base[k])
    acc = {a: [] for a in arms}
    for r in range(rounds + 1):
        for a in arms:
            p = pipes[a]
            for _ in range(30): p.step()
            p.results(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps): p.step()
            p.results(); torch.cuda.synchronize()
            if r: acc[a].append((time.perf_counter() - t0) / steps * 1e3)
for a, v in acc.items():
    print(f"{a:40s} {sorted(v)[len(v) // 2]:.4f} ms per step   [{', '.join(f'{x:.4f}' for x in v)}]", flush=True)
