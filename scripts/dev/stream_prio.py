"""Does stream priority / stagger change the overlap gain of 3 views on 3 streams (eager)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as Bn
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
B = 3
net, sd, feats, feats_cl, proj_d, dv_d, proj, dv = Bn.build_inputs(dev, 0, torch.float16, B)
def run(streams, n=60):
    main = torch.cuda.current_stream()
    def step():
        for b in range(B):
            streams[b].wait_stream(main)
            with torch.cuda.stream(streams[b]):
                net.hot_path([f[b:b + 1] for f in feats_cl], proj_d[b:b + 1], dv_d[b:b + 1])
        for b in range(B):
            main.wait_stream(streams[b])
    for _ in range(10): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
with torch.no_grad():
    for name, prios in [("all normal", (0, 0, 0)), ("all high", (-1, -1, -1)), ("one high", (-1, 0, 0)), ("two high", (-1, -1, 0)), ("all normal", (0, 0, 0))]:
        st = [torch.cuda.Stream(priority=p) for p in prios]
        t = run(st)
        print(f"{name:12s}: {t * 1e3:.3f} ms per 3-view step = {3 * Bn.VOX / t / 1e9:.2f} G voxels/s")
    # no per-step join: each stream runs its own views back to back (steps pipelined across the join)
    st = [torch.cuda.Stream() for _ in range(B)]
    n = 60
    def free_run():
        for b in range(B):
            with torch.cuda.stream(st[b]):
                for _ in range(n):
                    net.hot_path([f[b:b + 1] for f in feats_cl], proj_d[b:b + 1], dv_d[b:b + 1])
    free_run(); torch.cuda.synchronize(); t0 = time.perf_counter(); free_run(); torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / n
    print(f"no join between steps (upper bound of cross-step pipelining): {t * 1e3:.3f} ms per 3 views = {3 * Bn.VOX / t / 1e9:.2f} G voxels/s")
