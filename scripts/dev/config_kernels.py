#!/usr/bin/env python3
"""Dev: the full per-kernel table (HIP events of one eager forward) of a BASELINE configuration: python scripts/dev/config_kernels.py 5"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import run_configs as RC  # noqa: E402
from wild_deep_mvs_amd import ops, synthetic  # noqa: E402

from wild_deep_mvs_amd import _lib as L  # noqa: E402
for a_ in [x for x in sys.argv[1:] if "=" in x]:
    L.set_tuning(a_.split("=")[0], int(a_.split("=")[1]))
for cid in [int(x) for x in sys.argv[1:] if "=" not in x] or [5]:
    cfg = RC.CONFIGS[cid]
    net = RC.build(cfg["arch"])
    cfg["setup"](net)
    scene = synthetic.make_scene(1, cfg["V"], cfg["H"], cfg["W"], seed=cid)
    dev = {k: v.cuda() for k, v in scene.items()}
    call = lambda: net(dev["imgs"], dev["K"], dev["R"], dev["t"], dev["depth_min"], dev["depth_max"], **cfg["kw"])  # noqa: E731
    with torch.no_grad():
        net.graph_replay = False
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            with ops.EventTimer() as tm:
                call()
            d = tm.detail()
            if best is None or sum(v["ms"] for v in d.values()) < sum(v["ms"] for v in best.values()):
                best = d
    tot = sum(v["ms"] for v in best.values())
    import time
    with torch.no_grad():
        net.graph_replay = True
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); call(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"config {cid}: forward (graph replay) {sorted(ts)[3] * 1e3:.3f} ms")
    print(f"config {cid}: {tot:.3f} ms of engine kernels in {sum(v['launches'] for v in best.values())} launches")
    for r in RC.kernel_rooflines(best, top=60):
        print(f"  {r['kernel']:34s} x{r['launches']:3d}  avg {r['avg_us']:8.1f} us  total {r['avg_us'] * r['launches'] / 1e3:7.3f} ms  "
              f"hbm {0 if r['hbm_frac'] is None else r['hbm_frac']:.2f}  mfma {0 if r['mfma_frac'] is None else r['mfma_frac']:.2f}")
    del net, dev
    torch.cuda.empty_cache()
