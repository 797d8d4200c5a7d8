#!/usr/bin/env python3
"""Dev (round 4): how much would an LDS-staged group-correlation kernel buy on the Vis-MVSNet stage-1 shape (256 planes x 144 x 200, config 5)?
Proxy: on the SAME inputs (PROJ geometry, 4 source views of the 9-view rig) time the group-correlation launch (quad kernel, what runs today)
against the variance launch of the LDS-staged kernels (similar arithmetic per voxel-view: 128 blend FMAs + 64 against 128 + 32)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from wild_deep_mvs_amd import _lib as L, ops, synthetic  # noqa: E402


def time_us(run, steps=20, warm=5):
    for _ in range(warm):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


dev = torch.device("cuda", 0)
for a_ in sys.argv[1:]:
    if a_.startswith("--ppd="):
        L.set_tuning("warp_ppd", int(a_[6:]))
V, D, h, w = 9, 256, 144, 200
cm = synthetic.make_cameras(1, V, 8 * h, 8 * w)
Ks = cm["K"].clone()
Ks[:, :, :2] /= 8
proj = bench.build_proj_matrices(Ks, cm["R"], cm["t"]).to(dev)
steps = torch.arange(D, dtype=torch.float32).view(1, -1)
dv = (cm["depth_min"][:, :1] + (cm["depth_max"][:, :1] - cm["depth_min"][:, :1]) / (D - 1) * steps).to(dev).contiguous()
feats = synthetic.make_features(1, V, 32, h, w, seed=7)
fcl = [ops.to_channels_last(feats[i].to(dev), torch.float16) for i in range(V)]
import ctypes


def cam_array(K, R, t, start, interval):
    cam = torch.zeros((1, 2, 4, 4), device=dev)
    cam[:, 0, :3, :3], cam[:, 0, :3, 3:4], cam[:, 1, :3, :3] = R.to(dev), t.to(dev), K.to(dev)
    cam[:, 1, 3, 0], cam[:, 1, 3, 1] = float(start), float(interval)
    return cam


di = float((cm["depth_max"][0, 0] - cm["depth_min"][0, 0]) / 128)
arrs = [cam_array(cm["K"][:, i], cm["R"][:, i], cm["t"][:, i], cm["depth_min"][0, i], di) for i in range(V)]
for name, views in (("views 1-4", [1, 2, 3, 4]), ("views 5-8", [5, 6, 7, 8])):
    sub = proj[:, [0] + views].contiguous()
    cams_p = ops.proj_cams_device(sub.float().contiguous(), 0)
    cams_h = ops.homog_cams_device(arrs[0], [arrs[i] for i in views], 1.0 / 8)
    srcs = [fcl[i] for i in views]
    for label, tiled, cost, geom, cams in (("groupcorr HOMOG, quad kernel", 0, L.COST_GROUPCORR, L.GEOM_HOMOG, cams_h),
                                           ("groupcorr HOMOG, LDS-staged kernel (default)", 1, L.COST_GROUPCORR, L.GEOM_HOMOG, cams_h),
                                           ("variance PROJ, quad-owner LDS kernel", 1, L.COST_VARIANCE, L.GEOM_PROJ, cams_p),
                                           ("variance PROJ, lane-owner LDS kernel", 4, L.COST_VARIANCE, L.GEOM_PROJ, cams_p),
                                           ("variance PROJ, direct-gather quad kernel", 0, L.COST_VARIANCE, L.GEOM_PROJ, cams_p)):
        L.set_tuning("warp_tiled", tiled)
        hist = torch.zeros(16, dtype=torch.int32, device=dev)
        fn = L.lib().pscv_debug_wl_mode_hist
        fn.argtypes, fn.restype = [ctypes.c_void_p], None
        fn(hist.data_ptr())
        run = lambda: ops.warp_cost(fcl[0], srcs, cams, dv, cost=cost, geom=geom, out_dtype=torch.float16)  # noqa: E731
        run(); torch.cuda.synchronize(); fn(None)
        t = sorted(time_us(run) for _ in range(3))[1]
        L.set_tuning("warp_tiled", -1)
        hm = hist.view(4, 4).cpu().tolist()
        print(f"{name}: {label}: {t:.1f} us   modes [DIRECT, GEN, FAST, ZERO] per view {hm if sum(map(sum, hm)) else ''}", flush=True)
