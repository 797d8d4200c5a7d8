#!/usr/bin/env python3
"""ms per full forward (in-forward hipGraph replay) of BASELINE configurations for values of tuning knobs, arms interleaved.
Usage: python scripts/dev/knob_sweep_configs.py 4 default sweepc_slots=512 sweepc_slots=1024 ..."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import run_configs as RC
from wild_deep_mvs_amd import synthetic, _lib as L

cid = int(sys.argv[1])
arms = sys.argv[2:] or ["default"]
cfg = RC.CONFIGS[cid]
net = RC.build(cfg["arch"]); cfg["setup"](net)
scene = {k: v.cuda() for k, v in synthetic.make_scene(1, cfg["V"], cfg["H"], cfg["W"], seed=cid).items()}
call = lambda: net(scene["imgs"], scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"], **cfg["kw"])
knobs = sorted({kv.split("=")[0] for a in arms if a != "default" for kv in a.split(",")})
base = {k: L.get_tuning(k) for k in knobs}
acc = {a: [] for a in arms}
ref = None
with torch.no_grad():
    for r in range(3):
        for a in arms:
            for k in knobs: L.set_tuning(k, base[k])
            if a != "default":
                for kv in a.split(","):
                    k, v = kv.split("="); L.set_tuning(k, int(v))
            for _ in range(3): o = call()
            torch.cuda.synchronize()
            if ref is None: ref = o["depth"].clone()
            err = float((o["depth"] - ref).abs().max() / ref.abs().max())
            ts = []
            for _ in range(7):
                t0 = time.perf_counter(); call(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            if r: acc[a].append((sorted(ts)[3] * 1e3, err))
    for k in knobs: L.set_tuning(k, base[k])
for a, v in acc.items():
    print(f"config {cid} {a:32s} {min(x[0] for x in v):8.3f} ms (all: {', '.join(f'{x[0]:.3f}' for x in v)})  depth vs default arm: {max(x[1] for x in v):.1e}", flush=True)
