#!/usr/bin/env python3
"""The headline step (B reference views, one replayed hipGraph with one branch per view) under the scheduling knobs that decide
which kernels can share a CU: `warp_lds_pad` (KiB of LDS the LDS-staged warp kernel asks for on top of its 40: 0 -> four
workgroups per CU, 5 -> three, 14 -> two), lockstep against staggered branches (`MVSNet.batch_stagger`: view b + 1's warp waits
for view b's), B = 3 / 4, and two step graphs replayed alternately on two streams (step N + 1 may start under step N's tail).
Arms are interleaved over several rounds; medians.  Usage: python scripts/dev/step_matrix.py [--dtype bf16] [--rounds 3] [--arms ...]"""
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
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--tune", action="append", default=[])
    ap.add_argument("--arms", default="")
    args = ap.parse_args()
    for kv in args.tune:
        k, v = kv.split("=")
        L.set_tuning(k, int(v))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dt = Bn.DTYPES[args.dtype]
    inputs = {}

    def get_inputs(nb):
        if nb not in inputs:
            inputs[nb] = Bn.build_inputs(dev, 0, dt, nb)
        return inputs[nb]

    def capture(nb, stagger, pad, copies=1):
        net, sd, feats, fcl, proj_d, dv_d, _, _ = get_inputs(nb)
        net.batch_streams = True
        net.batch_stagger = bool(stagger)
        L.set_tuning("warp_lds_pad", pad)
        graphs = []
        with torch.no_grad():
            for _ in range(3):
                net.hot_path(fcl, proj_d, dv_d)
            torch.cuda.synchronize()
            for _ in range(copies):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    out = net.hot_path(fcl, proj_d, dv_d)
                graphs.append((g, out))
        L.set_tuning("warp_lds_pad", 0)
        net.batch_stagger = False
        return graphs

    arms = {}
    for nb in (3, 4):
        for pad in (0, 5, 14):
            for stg in (0, 1):
                arms[f"B{nb} pad{pad:02d} {'stagger ' if stg else 'lockstep'}"] = (nb, stg, pad, 1)
    arms["B3 pad00 lockstep, 2 graphs on 2 streams"] = (3, 0, 0, 2)
    arms["B3 pad00 stagger , 2 graphs on 2 streams"] = (3, 1, 0, 2)
    arms["B3 pad14 stagger , 2 graphs on 2 streams"] = (3, 1, 14, 2)
    arms["B3 pad05 stagger , 2 graphs on 2 streams"] = (3, 1, 5, 2)
    if args.arms:
        arms = {k: v for k, v in arms.items() if any(a in k for a in args.arms.split(","))}
    built = {k: capture(*v) for k, v in arms.items()}
    ref = {}
    for k, gl in built.items():     # results must not depend on the schedule
        nb = arms[k][0]
        for g, out in gl:
            g.replay()
        torch.cuda.synchronize()
        d = built[k][0][1][0].clone()
        if nb in ref:
            assert torch.equal(d, ref[nb]), f"{k}: depth differs from the first arm of B = {nb}"
        else:
            ref[nb] = d
    side = [torch.cuda.Stream(), torch.cuda.Stream()]
    samples = {k: [] for k in arms}

    def run(gl, n):
        if len(gl) == 1:
            g = gl[0][0]
            for _ in range(n):
                g.replay()
        else:
            main_s = torch.cuda.current_stream()
            for s in side:
                s.wait_stream(main_s)
            for i in range(n):
                with torch.cuda.stream(side[i & 1]):
                    gl[i & 1][0].replay()
            for s in side:
                main_s.wait_stream(s)

    for rnd in range(args.rounds + 1):
        for k, gl in built.items():
            run(gl, 30)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(gl, args.steps)
            torch.cuda.synchronize()
            dtm = (time.perf_counter() - t0) / args.steps
            if rnd:
                samples[k].append(dtm * 1e3)
    print(f"# headline step, {args.dtype} storage, {args.steps} replays per sample, {args.rounds} interleaved rounds; ms per step (median | all) and per view")
    for k, v in samples.items():
        nb = arms[k][0]
        med = sorted(v)[len(v) // 2]
        print(f"{k:48s} {med:7.4f} ms  = {med / nb * 1e3:6.1f} us per view  = {nb * Bn.VOX / med / 1e6:6.2f} G voxels/s   [{', '.join(f'{x:.4f}' for x in v)}]", flush=True)


if __name__ == "__main__":
    main()
