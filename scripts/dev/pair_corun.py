#!/usr/bin/env python3
"""Which kernels of the MVSNet hot path are COMPLEMENTARY on one MI355X?  For every pair (X, Y) of its stages -- warp + cost, conv0,
the eight small layers (conv1 ... conv9^T as one chain), the tail sweep, softargmin -- at the headline size: X alone (nx launches back
to back on stream A), Y alone (ny launches on stream B), then both at once (counts chosen so that the two streams are busy for about
the same time).  `gain` = (t_X + t_Y) / t_both: 1.0 = the GPU just time-slices them (no reason to co-schedule), 2.0 = perfectly
complementary.  Also: each stage beside ITSELF on two streams.  Usage: python scripts/dev/pair_corun.py [--dtype bf16] [--tune k=v]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench as Bn  # noqa: E402
from wild_deep_mvs_amd import _lib as L, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--tune", action="append", default=[])
    ap.add_argument("--target-us", type=float, default=2000.0)
    args = ap.parse_args()
    for kv in args.tune:
        k, v = kv.split("=")
        L.set_tuning(k, int(v))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dt = Bn.DTYPES[args.dtype]

    def make_stages(seed):
        net, sd, feats, fcl, proj_d, dv_d, _, _ = Bn.build_inputs(dev, seed, dt, 1)
        reg = net.cost_regularization
        ly = reg.engine_layers(dt)
        cams = ops.proj_cams_device(proj_d.float().contiguous(), 0)
        with torch.no_grad():
            cost = ops.warp_cost(fcl[0], fcl[1:], cams, dv_d, geom=L.GEOM_PROJ, cost=L.COST_VARIANCE, out_dtype=dt)
            c0 = ops.conv3d(cost, ly["conv0"])

            def small():
                c2 = ops.conv3d(ops.conv3d(c0, ly["conv1"]), ly["conv2"])
                c4 = ops.conv3d(ops.conv3d(c2, ly["conv3"]), ly["conv4"])
                c6 = ops.conv3d(ops.conv3d(c4, ly["conv5"]), ly["conv6"])
                u7 = ops.conv3d(c6, ly["conv7"], skip=c4)
                return ops.conv3d(u7, ly["conv9"], skip=c2)
            u9 = small()
            logits = ops.tail_sweep(u9, ly["conv11"], ly["prob"], skip=c0)
            dvf = dv_d.float().contiguous()
        return {
            "warp": lambda: ops.warp_cost(fcl[0], fcl[1:], cams, dv_d, geom=L.GEOM_PROJ, cost=L.COST_VARIANCE, out=cost),
            "conv0": lambda: ops.conv3d(cost, ly["conv0"], out=c0),
            "small8": small,
            "tail": lambda: ops.tail_sweep(u9, ly["conv11"], ly["prob"], skip=c0),
            "softargmin": lambda: ops.softargmin(logits, dvf, want_conf=True, conf_mode=0),
        }, (net, fcl, cost, c0, u9, logits)

    # two independent sets of tensors: stream A works on set 0, stream B on set 1 (no false sharing of outputs)
    S0, keep0 = make_stages(0)
    S1, keep1 = make_stages(1)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    names = list(S0)

    def graph_of(fn, n, stream):
        """n launches of fn captured on `stream` (replays cost one host call: no launch-rate limit in the measurement)"""
        g = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.stream(stream):
            fn()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
                for _ in range(n):
                    fn()
        return g

    def time_alone(g, stream, reps=5):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.cuda.stream(stream):
                g.replay()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best * 1e6

    def time_both(ga, gb, reps=5):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.cuda.stream(sa):
                ga.replay()
            with torch.cuda.stream(sb):
                gb.replay()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best * 1e6

    # single-launch times first (to size the repeat counts)
    one = {}
    for k in names:
        g = graph_of(S0[k], 20, sa)
        time_alone(g, sa, 2)
        one[k] = time_alone(g, sa) / 20
    print("# stage alone, us per launch (20 back to back in one graph): " + ", ".join(f"{k} {v:.1f}" for k, v in one.items()))
    count = {k: max(2, int(round(args.target_us / one[k]))) for k in names}
    GA = {k: graph_of(S0[k], count[k], sa) for k in names}
    GB = {k: graph_of(S1[k], count[k], sb) for k in names}
    TA = {k: time_alone(GA[k], sa) for k in names}
    TB = {k: time_alone(GB[k], sb) for k in names}
    print(f"# repeat counts for ~{args.target_us:.0f} us per stream: {count}")
    print(f"{'X (stream A)':12s} {'Y (stream B)':12s} {'t_X us':>9s} {'t_Y us':>9s} {'t_both us':>10s} {'gain':>6s}")
    for i, x in enumerate(names):
        for y in names[i:]:
            tb = time_both(GA[x], GB[y])
            print(f"{x:12s} {y:12s} {TA[x]:9.1f} {TB[y]:9.1f} {tb:10.1f} {(TA[x] + TB[y]) / tb:6.3f}", flush=True)


if __name__ == "__main__":
    main()
