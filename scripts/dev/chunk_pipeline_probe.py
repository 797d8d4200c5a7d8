#!/usr/bin/env python3
"""Dev (round 4), TIMING ONLY: would ONE view gain from pipelining its own depth halves -- warp[0:96) -> (conv0[0:96) beside
warp[96:192)) -> conv0[96:192) -- now that forked graphs replay correctly?  The boundary planes of the two conv0 launches are wrong
here (each pads with zeros at the cut); this measures time, not values.  Chunks: 2 and 3."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from wild_deep_mvs_amd import _lib as L, ops  # noqa: E402


def timeit(run, steps=60, warm=30):
    for _ in range(warm):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    dev = torch.device("cuda", 0)
    net, sd, feats, fcl, proj_d, dv_d, proj, dv = bench.build_inputs(dev, 0, torch.float16, 1)
    D = dv_d.shape[1]
    cams = ops.proj_cams_device(proj_d.float().contiguous(), 0)
    conv0 = net.cost_regularization.engine_layers(torch.float16)["conv0"]
    cost = torch.empty((1, D, bench.h, bench.w, 32), dtype=torch.float16, device=dev)
    c0 = torch.empty((1, D, bench.h, bench.w, 8), dtype=torch.float16, device=dev)
    side = [torch.cuda.Stream() for _ in range(4)]

    def seq():
        ops.warp_cost(fcl[0], fcl[1:], cams, dv_d, cost=L.COST_VARIANCE, out=cost)
        ops.conv3d(cost, conv0, out=c0)

    def piped(n):
        def run():
            main_s = torch.cuda.current_stream()
            cuts = [D * i // n // 8 * 8 for i in range(n + 1)]
            prev_warp_done = None
            for i in range(n):
                a, b = cuts[i], cuts[i + 1]
                st = side[i]
                st.wait_stream(main_s)
                if prev_warp_done is not None:
                    st.wait_event(prev_warp_done)              # stagger: chunk i's warp starts when chunk i-1's warp is done
                with torch.cuda.stream(st):
                    ops.warp_cost(fcl[0], fcl[1:], cams, dv_d[:, a:b].contiguous(), cost=L.COST_VARIANCE, out=cost[:, a:b])
                    ev = torch.cuda.Event(); ev.record(st)
                    prev_warp_done = ev
                    ops.conv3d(cost[:, a:b], conv0, out=c0[:, a:b])
            for i in range(n):
                main_s.wait_stream(side[i])
        return run

    with torch.no_grad():
        res = {}
        for name, fn in (("sequential", seq), ("2 chunks", piped(2)), ("3 chunks", piped(3)), ("4 chunks", piped(4))):
            fn(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            res[name] = timeit(g.replay) * 1e3
        print("warp + conv0 of ONE view, replayed graph, us: " + ", ".join(f"{k} {v:.1f}" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
