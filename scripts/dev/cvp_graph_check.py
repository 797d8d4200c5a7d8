#!/usr/bin/env python3
"""Eager vs eager vs hipGraph replay of the CVP-MVSNet forward (configuration 4 or a smaller size): per-level depth differences."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from run_configs import CONFIGS, build
from wild_deep_mvs_amd import synthetic
from wild_deep_mvs_amd.graph import GraphedModel
cfg = CONFIGS[4]
net = build(cfg["arch"]); cfg["setup"](net)
scene = synthetic.make_scene(1, cfg["V"], cfg["H"], cfg["W"], seed=4)
if "bscale" in cfg:
    scene["t"] = scene["t"] * cfg["bscale"]
dev = {k: v.cuda() for k, v in scene.items()}
call = lambda m: m(dev["imgs"], dev["K"], dev["R"], dev["t"], dev["depth_min"], dev["depth_max"], **cfg["kw"])
with torch.no_grad():
    a = call(net); b = call(net)
    g = GraphedModel(net)
    c = call(g); d = call(g)
keys = [k for k in a if isinstance(a[k], (list, tuple))]
print("keys", list(a.keys()))
def cmp(x, y, tag):
    if isinstance(x, (list, tuple)):
        for i, (u, v) in enumerate(zip(x, y)):
            cmp(u, v, f"{tag}[{i}]")
    elif torch.is_tensor(x):
        df = (x.float() - y.float()).abs()
        print(f"{tag}: max {float(df.max()):.4g} mean {float(df.mean()):.4g} frac>1e-3 {float((df > 1e-3).float().mean()):.4g}")
for k in a:
    cmp(a[k], b[k], f"eager-eager {k}")
for k in a:
    cmp(a[k], c[k], f"eager-graph1 {k}")
for k in a:
    cmp(c[k], d[k], f"graph1-graph2 {k}")
with torch.no_grad():
    for r in range(8):
        e = call(g)
        df = (e["depth"] - a["depth"]).abs()
        print(f"replay {r + 3}: depth max diff {float(df.max()):.4g}  frac>1e-3 {float((df > 1e-3).float().mean()):.3g}")
    # a second graph wrapper on the same model (what run_configs.time_config does after its eager loop)
    g2 = GraphedModel(net)
    for r in range(4):
        e = call(g2)
        df = (e["depth"] - a["depth"]).abs()
        print(f"second wrapper, call {r + 1}: depth max diff {float(df.max()):.4g}  frac>1e-3 {float((df > 1e-3).float().mean()):.3g}")
    x = call(net)
    print("eager again:", float((x["depth"] - a["depth"]).abs().max()))
