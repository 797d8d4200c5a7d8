#!/usr/bin/env python3
"""Where the 2.6 ms of the warp backward go (headline size, variance): a -DPSCV_ABLATE build switches off the global flush atomics
(1), the LDS atomics (2), the phase-A re-sampling (4).  bash scripts/dev/ab_build.sh abl "warp_bwd.hip" -DPSCV_ABLATE;
PSCV_LIB=$PWD/scripts/dev/libpscv_abl.so python scripts/dev/wbwd_ablate.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wild_deep_mvs_amd import _lib as L
if os.environ.get("PSCV_LIB"):
    L.LIB_PATH = os.environ["PSCV_LIB"]
from wild_deep_mvs_amd import ops, synthetic
from wild_deep_mvs_amd.models.MVSNet.model import build_proj_matrices

V, D, h, w, C = 5, 192, 128, 160, 32
dt = torch.bfloat16
cams = synthetic.make_cameras(1, V, 512, 640)
Ks = cams["K"].clone(); Ks[:, :, :2] /= 4
proj = build_proj_matrices(Ks, cams["R"], cams["t"]).cuda()
pc = ops.proj_cams_device(proj.float().contiguous(), 0)
g = torch.Generator().manual_seed(0)
feats = [(torch.randn(1, h, w, C, generator=g) * 0.5).to(dt).cuda() for _ in range(V)]
dv = torch.linspace(2.0, 6.0, D).view(1, D).cuda()
gcost = (torch.randn(1, D, h, w, C, generator=g) * 0.1).to(dt).cuda()

def run():
    return ops.warp_cost_bwd(feats[0], feats[1:], pc, dv, gcost, geom=L.GEOM_PROJ, cost=L.COST_VARIANCE)

def timeit(reps=10):
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for fl in (0, 64, 128, 64 + 47, 255 - 16, 255):
    L.set_tuning("fuse_c0", fl)
    print(f"flags {fl:3d} (1 no flush atomics, 2 no LDS atomics, 4 no phase A, 8 no phase-B sampling, 16 no bbox, 32 no flush scan, 64 no dref atomics, 128 no patch zeroing): {timeit():.3f} ms", flush=True)
L.set_tuning("fuse_c0", 0)
