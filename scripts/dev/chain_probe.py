#!/usr/bin/env python3
"""Free-running per-stream graphs with SEVERAL views chained in one graph: S streams x C views per graph (B = S x C views per step).
Does amortising the graph-launch gap of a stream over more views help?  us per view, interleaved arms.
Usage: python scripts/dev/chain_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as Bn

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
arms = [(3, 1), (3, 2), (3, 4), (2, 3), (4, 1), (4, 2)]
built = {}
with torch.no_grad():
    for S, C in arms:
        NB = S * C
        net, sd, feats, fcl, proj_d, dv_d, _, _ = Bn.build_inputs(dev, 0, torch.bfloat16, NB)
        streams = [torch.cuda.Stream() for _ in range(S)]
        graphs = []
        for s in range(S):
            def chain():
                outs = []
                for c in range(C):
                    b = s * C + c
                    outs.append(net.hot_path([f[b:b + 1] for f in fcl], proj_d[b:b + 1], dv_d[b:b + 1]))
                return outs
            with torch.cuda.stream(streams[s]):
                chain(); torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=streams[s], capture_error_mode="thread_local"):
                    out = chain()
            graphs.append((g, out))
        built[(S, C)] = (streams, graphs, (net, fcl, proj_d, dv_d))

def run(S, C, n):
    streams, graphs, _ = built[(S, C)]
    for _ in range(n):
        for st, (g, _) in zip(streams, graphs):
            with torch.cuda.stream(st):
                g.replay()

acc = {a: [] for a in arms}
for r in range(4):
    for S, C in arms:
        run(S, C, 20); torch.cuda.synchronize()
        n = max(20, 240 // (S * C))
        t0 = time.perf_counter(); run(S, C, n); torch.cuda.synchronize()
        if r: acc[(S, C)].append((time.perf_counter() - t0) / n / (S * C) * 1e6)
for (S, C), v in acc.items():
    print(f"{S} streams x {C} view(s) per graph (B = {S * C}): {sorted(v)[len(v) // 2]:6.1f} us per view   [{', '.join(f'{x:.1f}' for x in v)}]")
