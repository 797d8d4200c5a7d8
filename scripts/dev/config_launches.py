#!/usr/bin/env python3
"""Dev: every launch of one eager forward of a BASELINE configuration IN ORDER with its HIP-event duration (best of 3 passes by total):
python scripts/dev/config_launches.py 4 [knob=value ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import run_configs as RC
from wild_deep_mvs_amd import ops, synthetic, _lib as L
for a_ in [x for x in sys.argv[1:] if "=" in x]:
    L.set_tuning(a_.split("=")[0], int(a_.split("=")[1]))
for cid in [int(x) for x in sys.argv[1:] if "=" not in x] or [4]:
    cfg = RC.CONFIGS[cid]
    net = RC.build(cfg["arch"]); cfg["setup"](net)
    scene = synthetic.make_scene(1, cfg["V"], cfg["H"], cfg["W"], seed=cid)
    dev = {k: v.cuda() for k, v in scene.items()}
    call = lambda: net(dev["imgs"], dev["K"], dev["R"], dev["t"], dev["depth_min"], dev["depth_max"], **cfg["kw"])
    with torch.no_grad():
        net.graph_replay = False
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            with ops.EventTimer() as tm:
                call()
            torch.cuda.synchronize()
            rec = [(n, e0.elapsed_time(e1) * 1e3) for n, e0, e1 in tm.records]
            if best is None or sum(t for _, t in rec) < sum(t for _, t in best):
                best = rec
    print(f"config {cid}: {len(best)} launches, {sum(t for _, t in best) / 1e3:.3f} ms")
    acc = 0.0
    for i, (n, t) in enumerate(best):
        acc += t
        print(f"{i:4d} {n:36s} {t:9.1f} us   cum {acc / 1e3:7.3f} ms")
    del net, dev
    torch.cuda.empty_cache()
