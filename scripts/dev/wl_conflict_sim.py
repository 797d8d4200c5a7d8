"""Host simulation of the LDS-staged warp kernel's tap addresses on the bench scene: which lane groups of a `ds_read_b128` collide?
A lane group = 4 quads = 4 x-adjacent pixels of a tile row on one plane; each quad reads 64 contiguous bytes of a texel's lo (or
hi) plane; 256 B = 4 texel slots per LDS cycle, so two quads collide iff their texel indices are different but equal mod 4."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wild_deep_mvs_amd import synthetic
from wild_deep_mvs_amd.models.MVSNet.model import build_proj_matrices

V, D, h, w = 5, 192, 128, 160
cams = synthetic.make_cameras(1, V, 512, 640)
Ks = cams["K"].clone(); Ks[:, :, :2] /= 4
proj = build_proj_matrices(Ks, cams["R"], cams["t"])[0].double().numpy()
dv = np.linspace(2.0, 6.0, D)
ys, xs = np.mgrid[0:h, 0:w]
stats = {}
for v in range(1, V):
    P = proj[v] @ np.linalg.inv(proj[0])
    rot, tr = P[:3, :3], P[:3, 3]
    r = rot @ np.stack([xs.ravel(), ys.ravel(), np.ones(h * w)])           # [3, hw]
    tot = same_col_diff_row = span5 = other = groups = 0
    scale_x = []
    for d in dv[::4]:
        q = r * d + tr[:, None]
        u, vv = (q[0] / q[2]).reshape(h, w), (q[1] / q[2]).reshape(h, w)
        inside = (u >= 0) & (u <= w - 2) & (vv >= 0) & (vv <= h - 2)
        x0, y0 = np.floor(u).astype(int), np.floor(vv).astype(int)
        scale_x.append(np.median(np.diff(u, axis=1)))
        # groups of 4 x-adjacent pixels (columns 4k .. 4k+3); pitch is a multiple of 4 -> slot = x0 mod 4, texel id = (y0, x0)
        X = x0.reshape(h, w // 4, 4); Y = y0.reshape(h, w // 4, 4); ok = inside.reshape(h, w // 4, 4).all(-1)
        for tap_dx in (0, 1):                                  # the x0 and x0+1 taps are separate instructions
            Xs = X + tap_dx
            for i in range(4):
                for j in range(i + 1, 4):
                    coll = ok & ((Xs[..., i] - Xs[..., j]) % 4 == 0) & ((Xs[..., i] != Xs[..., j]) | (Y[..., i] != Y[..., j]))
                    tot += coll.sum()
                    same_col_diff_row += (coll & (Xs[..., i] == Xs[..., j])).sum()
                    span5 += (coll & (np.abs(Xs[..., i] - Xs[..., j]) == 4)).sum()
            groups += ok.sum()
    print(f"view {v}: median d(u)/dx {np.median(scale_x):.3f}; per lane-group instruction: colliding quad pairs {tot / groups:.3f} "
          f"(same texel column / other row {same_col_diff_row / groups:.3f}, columns 4 apart {span5 / groups:.3f})")
