#!/usr/bin/env python3
"""Phase timing of the regulariser's conv kernels with a -DPSCV_PROFILE build of libpscv (cycle stamps per workgroup).
Build:  bash scripts/dev/ab_build.sh prof "conv3d.hip conv3d_c1.hip conv3d_t2p8.hip" -DPSCV_PROFILE
Run:    PSCV_LIB=$PWD/scripts/dev/libpscv_prof.so python scripts/dev/phase_prof.py [layer ...]"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wild_deep_mvs_amd import _lib as L  # noqa: E402
L.LIB_PATH = os.environ["PSCV_LIB"]
from wild_deep_mvs_amd import ops  # noqa: E402

dev = "cuda"
dt = torch.float16
D, h, w = 192, 128, 160
g = torch.Generator().manual_seed(0)
# name, c_in, c_out, kind, input scale, transposed, skip?, getter tag
LAYERS = {
    "conv1": (8, 16, 1, 1, False, False, "s2s"),       # slots: prologue | stash + fetch issue | MFMA | epilogue | barrier
    "conv2": (16, 16, 0, 2, False, False, "conv"),
    "conv3": (16, 32, 1, 2, False, False, "conv"),
    "conv4": (32, 32, 0, 4, False, False, "conv"), "conv5": (32, 64, 1, 4, False, False, "conv"),
    "conv6": (64, 64, 0, 8, False, False, "conv"), "conv7": (64, 32, 2, 8, True, True, "conv"),
    "conv9": (32, 16, 2, 4, True, True, "conv"), "conv11": (16, 8, 2, 2, True, True, "t2p8"),
    "prob": (8, 1, 0, 1, False, False, "c1"),
    "conv0": (32, 8, 0, 1, False, False, "sweep"),     # slots: prologue | fetch issue | MFMA loop | epilogue | stash (waits for the planes) | barrier
    "cvp64": (64, 64, 0, (4, 512, 640), False, False, "wide"),
    "cvp16": (16, 16, 0, (8, 1024, 1280), False, False, "conv"),   # CVP conv0a at the finest level (brick kernel, 4x4x16 tiles)   # CVP refinement 64 -> 64 at 4 x 512 x 640 (4x4x16 tiles, all 4 N-tiles)
}
NAMES = ["loads issued", "loads landed", "LDS write+sync", "MFMA loop", "epilogue", "drain"]
lib = L.lib()
for name in sys.argv[1:] or list(LAYERS):
    ci, co, kind, s, tr, sk, tag = LAYERS[name]
    shp = s if isinstance(s, tuple) else (D // s, h // s, w // s)
    x = (torch.randn(1, *shp, ci, generator=g) * 0.5).to(dt).to(dev)
    wt = torch.randn(*((ci, co) if tr else (co, ci)), 3, 3, 3, generator=g) / (27 * ci) ** 0.5
    layer = ops.Conv3dLayer.build(wt, kind=kind, transposed=tr, device=dev, relu=co > 1, dtype=dt)
    Do, Ho, Wo = ops.conv_out_shape(kind, *x.shape[1:4])
    skip = (torch.randn(1, Do, Ho, Wo, co, generator=g) * 0.5).to(dt).to(dev) if sk else None
    out = torch.empty(1, Do, Ho, Wo, co, dtype=torch.float32 if co == 1 else dt, device=dev)
    fn = lambda: ops.conv3d(x, layer, skip=skip, out=out, out_dtype=out.dtype)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    time.sleep(0.05)          # the profiled launch stands alone in time: stale entries of earlier launches are > 1 M ticks older
    fn(); torch.cuda.synchronize()
    nb = 16384
    raw = (ctypes.c_uint * (nb * 16))()
    getattr(lib, "pscv_debug_prof_" + tag)(raw, nb)
    r = np.frombuffer(raw, dtype=np.uint32).reshape(nb, 16).astype(np.int64)
    # workgroups of the last launch.  Every XCD has its own cycle counter, so "newest" and the time origin are taken per XCD.
    xcc = r[:, 9] & 0xf
    valid = r[:, 7] != 0
    base_of = np.zeros(len(r), dtype=np.int64)
    for xc in np.unique(xcc[valid]):
        m = valid & (xcc == xc)
        newest = r[m, 7].max()
        keep = m & ((newest - r[:, 7]) % (1 << 32) < 1_000_000)
        valid &= ~m | keep
        base_of[keep] = r[keep, 6].min()
    r, xcc, base_of = r[valid], xcc[valid], base_of[valid]
    nblk = len(r)
    ph = r[:, :6]
    t0, t1 = r[:, 6] - base_of, r[:, 7] - base_of
    span = t1.max()
    key = ((xcc * 8 + ((r[:, 8] >> 13) & 7)) * 2 + ((r[:, 8] >> 12) & 1)) * 16 + ((r[:, 8] >> 8) & 0xf)
    ncu = len(np.unique(key))
    print(f"{name} ({ci}->{co}, kind {layer.kind}): {us:.1f} us  {nblk} workgroups profiled on {ncu} CUs / {len(np.unique(xcc))} XCDs, "
          f"launch span {span} ticks ({span / us:.0f} ticks/us if the span were the whole kernel)")
    tot = ph.sum(1)
    print("   mean ticks per workgroup (wave 0): total %7.0f | " % tot.mean() + "  ".join(f"{NAMES[i]} {ph[:, i].mean():6.0f}" for i in range(6)))
    print(f"   resident workgroups per CU (sum lifetimes / span / CUs): {(t1 - t0).sum() / span / ncu:.2f};  workgroup start times "
          f"p10 {np.percentile(t0, 10):.0f}  p50 {np.percentile(t0, 50):.0f}  p90 {np.percentile(t0, 90):.0f}  max {t0.max()};  end p50 "
          f"{np.percentile(t1, 50):.0f} max {t1.max()}")
    print('   wave slot (HW_ID & 15) histogram:', np.bincount(r[:, 8] & 0xf, minlength=16).tolist())
    for k in np.unique(key)[:2]:
        m = key == k
        o = np.argsort(t0[m])
        print(f"   CU {k}: " + " ".join(f"[{int(a_)}..{int(b_)}]" for a_, b_ in zip(t0[m][o][:14], t1[m][o][:14])))
