#!/usr/bin/env python3
"""CPU model of the LDS-staged warp kernel's box phase (csrc/warp_cost_tiled.hip): for both synthetic camera rigs at the headline size, the
share of (tile, plane range, source view) triples whose source box does NOT fit (-> global taps, mode DIRECT) as a function of the planes
per range (32 = whole chunk, 16 = halves, 8 = quarters), the widest box a staging wave covers and the arena size in texels; box sizes per view.
Reproduces the kernel's own histogram (bench.py alt_geometry: 0.43 / 0.15 on the DTU-like rig at 32 / 16 planes).  python scripts/dev/warp_box_sim.py"""
import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wild_deep_mvs_amd import synthetic
from wild_deep_mvs_amd.models.MVSNet.model import build_proj_matrices
V, H, W, D = 5, 512, 640, 192
h, w = H // 4, W // 4
for rig in ("probe", "dtu"):
    cams = synthetic.make_cameras(1, V, H, W, rig=rig)
    Ks = cams["K"].clone(); Ks[:, :, :2] /= 4
    P = build_proj_matrices(Ks, cams["R"], cams["t"])[0].double()      # [V,4,4]
    dmin, dmax = float(cams["depth_min"][0, 0]), float(cams["depth_max"][0, 0])
    dv = dmin + (dmax - dmin) / (D - 1) * np.arange(D)
    Pref_inv = torch.inverse(P[0])
    def boxes(nplanes, TW=8, TH=4):
        out = {}
        for v in range(1, V):
            M = (P[v] @ Pref_inv).numpy()
            rot, tr = M[:3, :3], M[:3, 3]
            res = []
            for d0 in range(0, D, nplanes):
                dl, dh = dv[d0], dv[min(D - 1, d0 + nplanes - 1)]
                for ty in range(0, h, TH):
                    for tx in range(0, w, TW):
                        us, vs, ok = [], [], True
                        for cx in (tx, min(tx + TW - 1, w - 1)):
                            for cy in (ty, min(ty + TH - 1, h - 1)):
                                for d in (dl, dh):
                                    q = rot @ np.array([cx, cy, 1.0]) * d + tr
                                    if q[2] <= 1e-6: ok = False
                                    us.append(q[0] / q[2]); vs.append(q[1] / q[2])
                        X0, X1 = int(np.floor(min(us) - 1 / 32)), int(np.floor(max(us) + 1 / 32)) + 1
                        Y0, Y1 = int(np.floor(min(vs) - 1 / 32)), int(np.floor(max(vs) + 1 / 32)) + 1
                        outside = X1 < 0 or Y1 < 0 or X0 > w - 1 or Y0 > h - 1
                        cX0, cX1, cY0, cY1 = max(X0, 0), min(X1, w - 1), max(Y0, 0), min(Y1, h - 1)
                        res.append((ok, outside, cX1 - cX0 + 1, cY1 - cY0 + 1))
            out[v] = res
        return out
    for npl in (32, 16, 8):
        b = boxes(npl)
        n = len(b[1])
        for maxw, arena in ((16, 318), (32, 318), (16, 400), (32, 400)):
            direct = np.zeros(V); zero = np.zeros(V)
            for i in range(n):
                used = 0
                for v in range(1, V):
                    ok, outside, bw, bh = b[v][i]
                    if not ok: direct[v] += 1; continue
                    if outside: zero[v] += 1; continue
                    pitch = (bw + 3) & ~3
                    if bw <= maxw and bh <= 8 and used + pitch * bh <= arena: used += pitch * bh
                    else: direct[v] += 1
            print(f"{rig} planes/range {npl:2d} maxw {maxw} arena {arena}: DIRECT share per view {np.round(direct[1:] / n, 3)} total {direct.sum() / (4 * n):.3f}  ZERO {zero.sum() / (4 * n):.3f}")
        ws = np.array([[x[2] for x in b[v] if x[0] and not x[1]] for v in range(1, V)], dtype=object)
        print("   box width mean/max per view:", [(round(float(np.mean(x)), 1), int(np.max(x))) for x in ws], " height:", [(round(float(np.mean([y[3] for y in b[v] if y[0] and not y[1]])), 1), int(np.max([y[3] for y in b[v] if y[0] and not y[1]]))) for v in range(1, V)])
