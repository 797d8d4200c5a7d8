#!/usr/bin/env python3
"""How the B views of a step should be handed to the GPU.  Arms (all the same launches, same results):
  forked   ONE hipGraph whose B views are parallel branches (round 3-5's step), replayed on one stream
  views    B single-branch hipGraphs (one per view), each replayed on its OWN stream; a step = one replay of each; the host never waits
           between steps (the streams run ahead of each other freely), one synchronize at the end of the timed region
  views+j  the same, but every step ends with a join on the caller's stream (step N + 1 starts after step N is complete)
Also prints the host time of a replay call (does hipGraphLaunch return before the GPU is done?).
Usage: python scripts/dev/view_graphs.py [--dtype bf16] [--batch 3] [--steps 200] [--rounds 3] [--mode run|trace]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench as Bn  # noqa: E402
from wild_deep_mvs_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=3)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--trace", default="", help="run only this arm for 40 steps (under rocprofv3 --kernel-trace)")
    ap.add_argument("--tune", action="append", default=[])
    args = ap.parse_args()
    for kv in args.tune:
        k, v = kv.split("=")
        L.set_tuning(k, int(v))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    NB = args.batch
    net, sd, feats, fcl, proj_d, dv_d, _, _ = Bn.build_inputs(dev, 0, Bn.DTYPES[args.dtype], NB)
    item = lambda b: net.hot_path([f[b:b + 1] for f in fcl], proj_d[b:b + 1], dv_d[b:b + 1])
    streams = [torch.cuda.Stream() for _ in range(NB)]
    with torch.no_grad():
        net.batch_streams = True
        for _ in range(3):
            net.hot_path(fcl, proj_d, dv_d)
        torch.cuda.synchronize()
        forked = torch.cuda.CUDAGraph()
        with torch.cuda.graph(forked, capture_error_mode="thread_local"):
            out_f = net.hot_path(fcl, proj_d, dv_d)
        views, outs = [], []
        for b in range(NB):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                outs.append(item(b))
            views.append(g)
        forked.replay()
        for g in views:
            g.replay()
        torch.cuda.synchronize()
        same = torch.equal(out_f[0], torch.cat([o[0] for o in outs], 0))
        print(f"# per-view graphs equal the forked graph bit for bit: {same}")

    main_s = torch.cuda.current_stream()

    def run_forked(n):
        for _ in range(n):
            forked.replay()

    def run_views(n, join):
        for s in streams:
            s.wait_stream(main_s)
        for _ in range(n):
            for b in range(NB):
                with torch.cuda.stream(streams[b]):
                    views[b].replay()
            if join:
                for s in streams:
                    main_s.wait_stream(s)
                for s in streams:
                    s.wait_stream(main_s)
        for s in streams:
            main_s.wait_stream(s)

    arms = {"forked": lambda n: run_forked(n), "views": lambda n: run_views(n, False), "views+j": lambda n: run_views(n, True)}
    if args.trace:
        arms[args.trace](40)
        torch.cuda.synchronize()
        return
    # host time of the launch calls
    for name, fn in arms.items():
        fn(10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(50)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"# {name:8s}: the host is back from 50 steps after {t_host * 1e3:7.2f} ms; the GPU is done after {t_all * 1e3:7.2f} ms")
    samples = {k: [] for k in arms}
    for rnd in range(args.rounds + 1):
        for name, fn in arms.items():
            fn(30)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn(args.steps)
            torch.cuda.synchronize()
            if rnd:
                samples[name].append((time.perf_counter() - t0) / args.steps * 1e3)
    for k, v in samples.items():
        med = sorted(v)[len(v) // 2]
        print(f"{k:10s} B={NB}: {med:7.4f} ms per step = {med / NB * 1e3:6.1f} us per view = {NB * Bn.VOX / med / 1e6:6.2f} G voxels/s   [{', '.join(f'{x:.4f}' for x in v)}]", flush=True)


if __name__ == "__main__":
    main()
