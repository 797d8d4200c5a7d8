#!/usr/bin/env python3
"""Where the weight-gradient kernel's time goes: a -DPSCV_ABLATE build (bash scripts/dev/ab_build.sh abl "conv3d_wgrad.hip warp_bwd.hip"
-DPSCV_ABLATE; PSCV_LIB=$PWD/scripts/dev/libpscv_abl.so) switches off the P staging (1), the Q loads (2), the Q LDS writes (4),
the MFMA loop (8) through pscv_set_tuning("fuse_c0", bits)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wild_deep_mvs_amd import _lib as L
if os.environ.get("PSCV_LIB"):
    L.LIB_PATH = os.environ["PSCV_LIB"]
from wild_deep_mvs_amd import ops

dt = torch.bfloat16
g = torch.Generator().manual_seed(0)
cases = [("MVSNet conv0 8x32 s1 192x128x160", 8, 32, 1, (1, 192, 128, 160)), ("Vis 8x8 s1 64x128x160", 8, 8, 1, (1, 64, 128, 160)),
         ("Vis 16x8 s2 32x64x80", 16, 8, 2, (1, 32, 64, 80)), ("CVP 64x64 s1 12x32x40", 64, 64, 1, (1, 12, 32, 40)),
         ("2-D 8x8 s1 5x1x512x640", 8, 8, 1, (5, 1, 512, 640))]

def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for name, ca, cb, s, (B, D, H, W) in cases:
    p = (torch.randn(B, D, H, W, ca, generator=g) * 0.1).to(dt).cuda()
    q = (torch.randn(B, D * s, H * s, W * s, cb, generator=g) * 0.5).to(dt).cuda()
    row = []
    for fl in (0, 1, 2, 4, 6, 8, 15):
        L.set_tuning("fuse_c0", fl)
        row.append(f"{fl}:{timeit(lambda: ops.conv3d_wgrad(p, q, ca=ca, cb=cb, stride=s)):.0f}")
    L.set_tuning("fuse_c0", 0)
    print(f"{name}: us by flags (1 no P staging, 2 no Q loads, 4 no Q LDS writes, 8 no MFMA)  " + "  ".join(row), flush=True)
