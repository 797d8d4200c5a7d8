"""K independent reference views in flight on K streams (fork / join inside one captured graph): throughput per view."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as Bn
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
net, sd, feats, feats_cl, proj_d, dv_d, proj, dv = Bn.build_inputs(dev, 0, torch.float16)
sets = [[torch.roll(f, shifts=(3 * k, 2 * k), dims=(1, 2)).contiguous() for f in feats_cl] for k in range(6)]
with torch.no_grad():
    refs = [net.hot_path(s, proj_d, dv_d)[0].clone() for s in sets]
    torch.cuda.synchronize()
    for nv, ns in [(1, 1), (2, 1), (2, 2), (3, 3), (4, 2), (4, 4), (6, 2), (6, 3)]:
        g = torch.cuda.CUDAGraph()
        side = [torch.cuda.Stream() for _ in range(ns)]
        with torch.cuda.graph(g):
            main = torch.cuda.current_stream()
            outs = [None] * nv
            for st in side:
                st.wait_stream(main)
            for i in range(nv):
                with torch.cuda.stream(side[i % ns]):
                    outs[i] = net.hot_path(sets[i], proj_d, dv_d)
            for st in side:
                main.wait_stream(st)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 120
        for _ in range(n):
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        ok = all(bool(torch.equal(outs[i][0], refs[i])) for i in range(nv))
        print(f"{nv} views on {ns} stream(s): {dt * 1e6:7.1f} us per replay = {dt / nv * 1e6:6.1f} us per view = {nv * Bn.VOX / dt / 1e9:5.2f} G voxels/s, exact: {ok}")
