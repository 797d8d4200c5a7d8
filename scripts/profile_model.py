#!/usr/bin/env python3
"""Where does a full forward() spend its GPU time?  pscv launches are bracketed by HIP events (ops.EventTimer); the
rest (2-D feature nets and other PyTorch-ROCm work, launch gaps) is the difference to the wall time.
Usage: python scripts/profile_model.py --config 3"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from run_configs import CONFIGS, build  # noqa: E402
from wild_deep_mvs_amd import ops, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    net = build(cfg["arch"])
    cfg["setup"](net)
    scene = synthetic.make_scene(1, cfg["V"], cfg["H"], cfg["W"], seed=args.config)
    if "bscale" in cfg:
        scene["t"] = scene["t"] * cfg["bscale"]
    dev = {k: v.cuda() for k, v in scene.items()}
    call = lambda: net(dev["imgs"], dev["K"], dev["R"], dev["t"], dev["depth_min"], dev["depth_max"], **cfg["kw"])
    for _ in range(2):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    call()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    with ops.EventTimer() as tm:
        call()
    summ = tm.summary()
    tot = sum(ms for _, ms in summ.values())
    print(f"config {args.config}: wall {wall * 1e3:.2f} ms, pscv kernels {tot:.2f} ms in {sum(n for n, _ in summ.values())} launches")
    for k, (n, ms) in sorted(summ.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"   {k:28s} x{n:4d}  {ms:8.3f} ms")
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        call()
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))


if __name__ == "__main__":
    main()
