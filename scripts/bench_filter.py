#!/usr/bin/env python3
"""Geometric-consistency filter at evaluation size: pscv_geo_filter on the GPU, the oracle (= the
reference's CPU tensor chain) on the host cores.  Prints one JSON line.
Usage: python scripts/bench_filter.py [--h 1152 --w 1600 --views 11 --reps 50] [--no-cpu]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wild_deep_mvs_amd import ops, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--h", type=int, default=1152)
    ap.add_argument("--w", type=int, default=1600)
    ap.add_argument("--views", type=int, default=11)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    sc = synthetic.make_filter_scene(a.views, a.h, a.w, seed=0)
    n = a.views - 1
    d, src = sc["depth"].cuda(), [s.cuda() for s in sc["src_depth"]]
    cams = ops.geo_filter_cams(sc["K"], sc["R"], sc["t"]).cuda()
    for _ in range(3):
        ops.geo_filter(d, src, cams)
    torch.cuda.synchronize()
    s = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            for _ in range(10):
                ops.geo_filter(d, src, cams)
        g.replay(); torch.cuda.synchronize()
        e0.record(side)
        for _ in range(a.reps // 10):
            g.replay()
        e1.record(side)
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / (a.reps // 10 * 10) * 1e3
    alg = (a.views * a.h * a.w * 4 + 3 * a.h * a.w)
    out = {"metric": "geometric-consistency filter, reference pixels x source views / s", "value": a.h * a.w * n / us * 1e6,
           "unit": "pixel-views/s", "us_per_image": us, "config": {"h": a.h, "w": a.w, "src_views": n},
           "roofline": {"bound": "hbm", "achieved": alg / us / 1e3, "peak": 8000.0, "unit": "GB/s", "frac": alg / us / 1e3 / 8000.0,
                        "algorithmic_bytes": alg}}
    if not a.no_cpu:
        from oracle import filtering as OF
        torch.set_num_threads(os.cpu_count())
        t0 = time.time()
        OF.geometric_masks(sc["depth"], sc["src_depth"], sc["K"], sc["R"], sc["t"])
        dt = time.time() - t0
        out["cpu_baseline"] = {"value": a.h * a.w * n / dt, "unit": "pixel-views/s", "cores": os.cpu_count(), "kind": "port",
                               "sample": f"1 image, {n} source views, fp32, {dt:.2f} s"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
