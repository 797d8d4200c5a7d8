"""Is a multi-branch captured graph replayed correctly on this ROCm?  MVSNet hot path, B items on B streams; inputs CHANGE before every
replay (stale reads must show); variants: (A) torch.cat of the items' results behind the join, (B) no node behind the join."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as Bn
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
net, sd, feats, feats_cl, proj_d, dv_d, proj, dv = Bn.build_inputs(dev, 0, torch.float16, B)
static = [f.clone() for f in feats_cl]
gen = torch.Generator(device="cuda").manual_seed(0)


def eager(fs):
    net.batch_streams = False
    outs = [net.hot_path([f[b:b + 1] for f in fs], proj_d[b:b + 1], dv_d[b:b + 1]) for b in range(B)]
    net.batch_streams = True
    return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])


with torch.no_grad():
    for _ in range(2):
        net.hot_path(static, proj_d, dv_d)
    torch.cuda.synchronize()
    gA = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gA):
        outA = net.hot_path(static, proj_d, dv_d)
    # variant B: branches only, results stay per item
    gB = torch.cuda.CUDAGraph()
    side = [torch.cuda.Stream() for _ in range(B)]
    with torch.cuda.graph(gB):
        main = torch.cuda.current_stream()
        outB = [None] * B
        for b in range(B):
            side[b].wait_stream(main)
            with torch.cuda.stream(side[b]):
                outB[b] = net.hot_path([f[b:b + 1] for f in static], proj_d[b:b + 1], dv_d[b:b + 1])
        for b in range(B):
            main.wait_stream(side[b])
    bad = {"A": 0, "B": 0}
    worst = {"A": 0.0, "B": 0.0}
    for rep in range(12):
        for f in static:
            f.copy_((torch.randn(f.shape, generator=gen, device="cuda") * 0.5).to(f.dtype))
        ref_d, ref_c = eager(static)
        torch.cuda.synchronize()
        gA.replay(); torch.cuda.synchronize()
        eA = float((outA[0] - ref_d).abs().max())
        gB.replay(); torch.cuda.synchronize()
        eB = max(float((outB[b][0] - ref_d[b:b + 1]).abs().max()) for b in range(B))
        bad["A"] += eA > 0; bad["B"] += eB > 0
        worst["A"] = max(worst["A"], eA); worst["B"] = max(worst["B"], eB)
    print(f"B={B}: variant A (cat behind the join): {bad['A']} of 12 replays differ (max abs {worst['A']:.3e}); "
          f"variant B (nothing behind the join): {bad['B']} of 12 differ (max abs {worst['B']:.3e})")
