#!/usr/bin/env python3
"""How the LDS-staged warp kernel's time depends on the workgroups a CU can hold: a -DPSCV_ABLATE build asks for k KiB of LDS on top
of its 40 KiB (pscv_set_tuning("fuse_c0", k)): 0 -> 4 per CU (3 by the runtime's count), 14 -> 2, 42 -> 1.  Headline size, f16.
bash scripts/dev/ab_build.sh abl "warp_cost_tiled.hip" -DPSCV_ABLATE; PSCV_LIB=$PWD/scripts/dev/libpscv_abl.so python scripts/dev/wl_residency.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wild_deep_mvs_amd import _lib as L
if os.environ.get("PSCV_LIB"):
    L.LIB_PATH = os.environ["PSCV_LIB"]
from wild_deep_mvs_amd import ops, synthetic
from wild_deep_mvs_amd.models.MVSNet.model import build_proj_matrices

V, D, h, w, C = 5, 192, 128, 160, 32
dt = torch.float16
cams = synthetic.make_cameras(1, V, 512, 640)
Ks = cams["K"].clone(); Ks[:, :, :2] /= 4
pc = ops.proj_cams_device(build_proj_matrices(Ks, cams["R"], cams["t"]).cuda().float().contiguous(), 0)
g = torch.Generator().manual_seed(0)
feats = [(torch.randn(1, h, w, C, generator=g) * 0.5).to(dt).cuda() for _ in range(V)]
dv = torch.linspace(2.0, 6.0, D).view(1, D).cuda()
out = torch.empty(1, D, h, w, C, dtype=dt, device="cuda")

def timeit(reps=30):
    f = lambda: ops.warp_cost(feats[0], feats[1:], pc, dv, geom=L.GEOM_PROJ, cost=L.COST_VARIANCE, out=out)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for extra in (0, 1, 14, 27, 42, 82):
    L.set_tuning("fuse_c0", extra)
    lds = 40.3 + extra
    print(f"+{extra:2d} KiB LDS ({int(160 // lds)} workgroups per CU by LDS): {timeit():7.1f} us", flush=True)
L.set_tuning("fuse_c0", 0)
