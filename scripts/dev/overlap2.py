"""Experiment: two independent reference views in flight on two streams (view B's VALU-bound warp beside view A's MFMA / memory-bound
U-Net) against one after the other -- throughput of the headline hot path per GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as Bn
from wild_deep_mvs_amd import ops

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
net, sd, feats, feats_cl, proj_d, dv_d, proj, dv = Bn.build_inputs(dev, 0, torch.float16)
feats2 = [torch.roll(f, shifts=(7, 5), dims=(1, 2)).contiguous() for f in feats_cl]      # a DIFFERENT view set: races must show
K = 200


def run_graph(g, n):
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


with torch.no_grad():
    for _ in range(3):
        net.hot_path(feats_cl, proj_d, dv_d)
    torch.cuda.synchronize()
    # (1) one view per replay
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        d1, c1 = net.hot_path(feats_cl, proj_d, dv_d)
    t1 = run_graph(g1, K)
    d2, c2 = net.hot_path(feats2, proj_d, dv_d)
    torch.cuda.synchronize()
    # eager, two streams, staggered: is it the graph or the kernels?
    s_e = torch.cuda.Stream()
    for rep in range(3):
        cams = ops.proj_cams_device(proj_d.to(torch.float32).contiguous(), 0)
        costA = net.build_cost_volume(feats_cl[0], feats_cl[1:], proj_d[:, 0], [proj_d[:, i] for i in range(1, 5)], dv_d, cams)
        s_e.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_e):
            be = net.hot_path(feats2, proj_d, dv_d)
        la, oa = net.cost_regularization(costA, None, regress=dv_d)
        torch.cuda.current_stream().wait_stream(s_e)
        torch.cuda.synchronize()
        print(f"eager two streams staggered: a max abs {float((oa['depth'] - d1).abs().max()):.3e}, b {float((be[0] - d2).abs().max()):.3e}")
    # (2) two views per replay, one after the other on one stream
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        a = net.hot_path(feats_cl, proj_d, dv_d)
        b = net.hot_path(feats2, proj_d, dv_d)
    t2 = run_graph(g2, K // 2)
    # (3) two views per replay on two streams (fork / join inside the capture), the second one delayed by nothing: the hardware
    #     queues decide the mix
    for stagger in (False, True):
        g3 = torch.cuda.CUDAGraph()
        s_b = torch.cuda.Stream()
        with torch.cuda.graph(g3):
            main = torch.cuda.current_stream()
            s_b.wait_stream(main)
            if stagger:
                # view A's warp first, then B's warp beside A's U-Net: A = warp on main; B's stream waits for A's warp
                cams = ops.proj_cams_device(proj_d.to(torch.float32).contiguous(), 0)
                costA = net.build_cost_volume(feats_cl[0], feats_cl[1:], proj_d[:, 0], [proj_d[:, i] for i in range(1, 5)], dv_d, cams)
                s_b.wait_stream(main)
                with torch.cuda.stream(s_b):
                    b = net.hot_path(feats2, proj_d, dv_d)
                la, oa = net.cost_regularization(costA, None, regress=dv_d)
                a = (oa["depth"], oa["conf"])
            else:
                with torch.cuda.stream(s_b):
                    b = net.hot_path(feats2, proj_d, dv_d)
                a = net.hot_path(feats_cl, proj_d, dv_d)
            main.wait_stream(s_b)
        t3 = run_graph(g3, K // 2)
        ok = bool(torch.equal(a[0], d1)) and bool(torch.equal(b[0], d2))
        print(f"two streams (stagger={stagger}): {t3 * 1e6:.1f} us per 2 views = {2 * Bn.VOX / t3 / 1e9:.2f} G voxels/s; outputs equal the single-view run: {ok}"
              f" (a: max abs {float((a[0] - d1).abs().max()):.3e}, conf {float((a[1] - c1).abs().max()):.3e}; b: {float((b[0] - d2).abs().max()):.3e}, conf {float((b[1] - c2).abs().max()):.3e})")
    print(f"one view per replay: {t1 * 1e6:.1f} us = {Bn.VOX / t1 / 1e9:.2f} G voxels/s")
    print(f"two views, one stream: {t2 * 1e6:.1f} us per 2 views = {2 * Bn.VOX / t2 / 1e9:.2f} G voxels/s")
