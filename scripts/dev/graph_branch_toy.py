"""Toy: does a captured two-branch graph keep intra-branch order on this ROCm?  Each branch: y = x + 1; z = y * 2; w = z - 3 (separate
kernels, fresh allocations); inputs change before every replay.  And: eager two-stream throughput of the MVSNet hot path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.cuda.set_device(0)
N = 1 << 24
xs = [torch.zeros(N, device="cuda") for _ in range(2)]
side = [torch.cuda.Stream() for _ in range(2)]
def chain(x):
    y = x + 1
    for _ in range(6):
        y = y * 1.0001 + 0.5
    z = y * 2
    return z - 3
for x in xs: chain(x)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    main = torch.cuda.current_stream()
    outs = [None, None]
    for b in range(2):
        side[b].wait_stream(main)
        with torch.cuda.stream(side[b]):
            outs[b] = chain(xs[b])
    for b in range(2):
        main.wait_stream(side[b])
bad = 0
for rep in range(10):
    for b in range(2):
        xs[b].fill_(float(rep * 2 + b))
    g.replay(); torch.cuda.synchronize()
    for b in range(2):
        bad += not torch.equal(outs[b], chain(xs[b]))
print("toy two-branch graph: wrong replays:", bad, "of 20")

import bench as Bn
dev = torch.device("cuda", 0)
net, sd, feats, feats_cl, proj_d, dv_d, proj, dv = Bn.build_inputs(dev, 0, torch.float16, 2)
with torch.no_grad():
    for flag in (False, True, False, True):
        net.batch_streams = flag
        for _ in range(5):
            net.hot_path(feats_cl, proj_d, dv_d)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            net.hot_path(feats_cl, proj_d, dv_d)
        torch.cuda.synchronize()
        print(f"eager, batch of 2, streams={flag}: {(time.perf_counter() - t0) / 100 * 1e3:.3f} ms per step")
