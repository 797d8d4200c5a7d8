#!/usr/bin/env python3
"""Dev (round 4): the 3-view headline step as (A) one batched launch per layer in a replayed hipGraph (the bench default), (B) one
HIP stream per view, eager, (C) one stream per view FORKED INSIDE a captured hipGraph (round 3 saw such graphs replay wrongly and
blamed ROCm; the cause was the packed warp build, so this is re-tested with the scalar build: replays on CHANGING inputs against the
eager result), each with the warp kernel at 4 / 3 / 2 workgroups per CU ("warp_lds_pad" = 0 / 12 / 40 KiB: room for another view's
conv0 on the same CU)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from wild_deep_mvs_amd import _lib as L, ops, synthetic  # noqa: E402


def timeit(run, steps=40, warm=40):
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
    NB = 3
    net, sd, feats, feats_cl, proj_d, dv_d, proj, dv = bench.build_inputs(dev, 0, torch.float16, NB)
    alt = [ops.to_channels_last((feats[i] * 0.9 + 0.05).to(dev), torch.float16) for i in range(bench.V)]       # a second input set
    with torch.no_grad():
        net.batch_streams = False
        want = [tuple(t.clone() for t in net.hot_path(f, proj_d, dv_d)) for f in (feats_cl, alt)]
        for pad in (0, 12, 40):
            L.set_tuning("warp_lds_pad", pad)
            res = {}
            # (A) batched, graph
            net.batch_streams = False
            static = [f.clone() for f in feats_cl]
            g = torch.cuda.CUDAGraph()
            net.hot_path(static, proj_d, dv_d); torch.cuda.synchronize()
            with torch.cuda.graph(g):
                out = net.hot_path(static, proj_d, dv_d)
            res["A batched graph"] = timeit(g.replay)
            okA = torch.equal(out[0], want[0][0])
            # (B) streams, eager
            net.batch_streams = True
            res["B streams eager"] = timeit(lambda: net.hot_path(feats_cl, proj_d, dv_d))
            o = net.hot_path(feats_cl, proj_d, dv_d); torch.cuda.synchronize()
            okB = torch.equal(o[0], want[0][0])
            # (C) streams forked inside a captured graph, replayed on changing inputs
            net.batch_streams_capture = True
            okC, tC = None, None
            try:
                static = [f.clone() for f in feats_cl]
                net.hot_path(static, proj_d, dv_d); torch.cuda.synchronize()
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2):
                    out2 = net.hot_path(static, proj_d, dv_d)
                bad = 0
                for it in range(12):
                    src = (feats_cl, alt)[it % 2]
                    for s_, f in zip(static, src):
                        s_.copy_(f)
                    g2.replay(); torch.cuda.synchronize()
                    bad += not (torch.equal(out2[0], want[it % 2][0]) and torch.equal(out2[1], want[it % 2][1]))
                okC = bad
                tC = timeit(g2.replay)
            except Exception as e:
                okC = f"capture failed: {type(e).__name__}: {e}"[:200]
            finally:
                net.batch_streams_capture = False
            res["C streams in graph"] = tC
            print(f"warp_lds_pad={pad:2d} KiB: " + ", ".join(f"{k} {v:.3f} ms" if v is not None else f"{k} n/a" for k, v in res.items()) +
                  f" | A equals reference: {okA}, B equals: {okB}, C wrong replays of 12 on changing inputs: {okC}", flush=True)
        L.set_tuning("warp_lds_pad", 0)
        for ppd in (16, 24, 32, 48):
            L.set_tuning("warp_ppd", ppd)
            fcl1 = [f[:1].contiguous() for f in feats_cl]
            w = lambda: net.build_cost_volume(fcl1[0], fcl1[1:], None, None, dv_d[:1], ops.proj_cams_device(proj_d[:1].float().contiguous(), 0))
            cams = ops.proj_cams_device(proj_d[:1].float().contiguous(), 0)
            w = lambda: ops.warp_cost(fcl1[0], fcl1[1:], cams, dv_d[:1].contiguous(), cost=L.COST_VARIANCE, out_dtype=torch.float16)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(5):
                w()
            e0.record()
            for _ in range(30):
                w()
            e1.record(); torch.cuda.synchronize()
            print(f"warp_ppd={ppd}: warp + cost alone {e0.elapsed_time(e1) / 30 * 1e3:.1f} us", flush=True)
        L.set_tuning("warp_ppd", 0)


if __name__ == "__main__":
    main()
