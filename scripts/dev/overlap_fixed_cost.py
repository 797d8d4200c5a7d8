"""Where does the fixed cost of a timed region of multi-branch graph replays sit?  n replays between two synchronisations, n = 1..64."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as Bn
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
net, sd, feats, feats_cl, proj_d, dv_d, proj, dv = Bn.build_inputs(dev, 0, torch.float16, 2)
with torch.no_grad():
    for _ in range(3):
        net.hot_path(feats_cl, proj_d, dv_d)
    torch.cuda.synchronize()
    graphs = {}
    for name, flag in (("two streams", True), ("one stream (batched launches)", False)):
        net.batch_streams = flag
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            net.hot_path(feats_cl, proj_d, dv_d)
        graphs[name] = g
    for name, g in graphs.items():
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize()
        row = []
        for n in (1, 2, 4, 8, 16, 32, 64):
            best = 1e9
            for rep in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    g.replay()
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            row.append(f"n={n}: {best * 1e3:.3f} ms ({best / n * 1e3:.3f}/replay)")
        print(name, "|", "; ".join(row))
    # host cost of a replay call
    g = graphs["two streams"]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(64):
        g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"host time of 64 replay() calls (two streams): {(t1 - t0) * 1e3:.3f} ms; total incl. sync {(time.perf_counter() - t0) * 1e3:.3f} ms")
