"""Kernel trace of N graph-replayed forwards of one BASELINE configuration, every kernel listed (torch's too).
usage (on the GPU box): python scripts/dev/trace_cfg.py <cfg> [n]   -- run under rocprofv3 by trace_cfg.sh"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import run_configs as RC
from wild_deep_mvs_amd import synthetic
cid = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = RC.CONFIGS[cid]
net = RC.build(cfg["arch"]); cfg["setup"](net)
net.graph_replay = len(sys.argv) <= 3
scene = synthetic.make_scene(1, cfg["V"], cfg["H"], cfg["W"], seed=cid)
if "bscale" in cfg:
    scene["t"] = scene["t"] * cfg["bscale"]
dev = {k: v.cuda() for k, v in scene.items()}
with torch.no_grad():
    for _ in range(n + 2):
        net(dev["imgs"], dev["K"], dev["R"], dev["t"], dev["depth_min"], dev["depth_max"], **cfg["kw"])
        torch.cuda.synchronize()
        time.sleep(0.25)          # the trace is split into forwards at these gaps
