#!/usr/bin/env python3
"""Phase timing of the LDS-staged warp kernel with a -DWL_PROFILE build of libpscv (cycle stamps summed per block).
Build: hipcc -DWL_PROFILE -c warp_cost_tiled.hip, link as libpscv_prof.so, point PSCV_LIB at it."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wild_deep_mvs_amd import _lib as L
L.LIB_PATH = os.environ["PSCV_LIB"]
from wild_deep_mvs_amd import ops, synthetic
from wild_deep_mvs_amd.models.MVSNet.model import build_proj_matrices
dev = "cuda"; V, D, h, w = 5, 192, 128, 160
feats = synthetic.make_features(1, V, 32, h, w, seed=1)
fcl = [ops.to_channels_last(feats[i].to(dev), torch.float16) for i in range(V)]
cams = synthetic.make_cameras(1, V, 512, 640)
Ks = cams["K"].clone(); Ks[:, :, :2] /= 4
proj = build_proj_matrices(Ks, cams["R"], cams["t"]).to(dev)
dv = torch.linspace(2.0, 6.0, D).view(1, D).to(dev)
cm = ops.proj_cams([proj[:, i] for i in range(1, V)], proj[:, 0])
out = torch.empty(1, D, h, w, 32, dtype=torch.float16, device=dev)
lib = L.lib()
TEMP = float(os.environ.get("WL_TEMP", "1.0"))
L.set_tuning("warp_tile", int(os.environ.get("WL_VARIANT", "0")))
TH = int(os.environ.get("WL_TILE_H", "4"))
for ppd in [int(x) for x in sys.argv[1:]] or [32]:
    L.set_tuning("warp_tiled", 1); L.set_tuning("warp_ppd", ppd)
    buf = (ctypes.c_ulonglong * 16)()
    for _ in range(3):
        ops.warp_cost(fcl[0], fcl[1:], cm, dv, cost=L.COST_VARIANCE, out=out, temp=TEMP)
    torch.cuda.synchronize()
    nblk = (D + ppd - 1) // ppd * (h // TH) * (w // 8)
    lib.pscv_debug_wl_prof(buf, nblk)
    names = ["lane consts issued", "box phase (wave 0) / wait", "barrier 1", "table read + fill", "barrier 2", "sweep"]
    print(f"ppd={ppd}: average cycles per block")
    for wv, off in (("wave 0", 0), ("last wave", 8)):
        tot = sum(buf[off + i] for i in range(6)) / nblk
        print(f"  {wv}: total {tot:8.0f}  " + "  ".join(f"{names[i]} {buf[off + i] / nblk:7.0f}" for i in range(6)))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.warp_cost(fcl[0], fcl[1:], cm, dv, cost=L.COST_VARIANCE, out=out, temp=TEMP)
    e1.record(); torch.cuda.synchronize()
    print(f"  kernel time {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
    # per-CU timelines from the absolute stamps of the last launch
    import numpy as np
    raw = (ctypes.c_uint * (nblk * 16))()
    lib.pscv_debug_wl_raw(raw, nblk)
    r = np.frombuffer(raw, dtype=np.uint32).reshape(nblk, 16).astype(np.int64)
    t0b, t1b, t1w7 = r[:, 6], r[:, 7], r[:, 8 + 7]
    hwid, xcc = r[:, 0], r[:, 8]
    cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    base = t0b.min()
    span = (np.maximum(t1b, t1w7) - base).max()
    print(f"  launch span {span} ticks; distinct CU keys {len(np.unique(key))}")
    tot = 0
    for k in np.unique(key):
        m = key == k
        tot += (np.maximum(t1b[m], t1w7[m]) - t0b[m]).sum()
    print(f'  mean resident blocks per CU over the span: {tot / span / len(np.unique(key)):.2f}')
    for k in np.unique(key)[:3]:
        m = key == k
        o = np.argsort(t0b[m])
        print(f"  CU key {k}: " + " ".join(f"[{int(a - base)}..{int(max(b_, c_) - base)}]" for a, b_, c_ in zip(t0b[m][o], t1b[m][o], t1w7[m][o])))
