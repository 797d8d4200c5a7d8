#!/usr/bin/env python3
"""Time the unsupervised photometric loss forward + backward on one MI355X: the pscv kernels
(``models.trainer.Trainer.photometricloss``) beside the same loss written with ATen ops the way the reference runs it
(``F.grid_sample`` per view + five grouped ``F.conv2d`` per SSIM; models/trainer.py:221-238, utils/ssimLoss.py:27-44).

    python scripts/bench_photo.py [--views 5 --height 128 --width 160 --batch 1 --iters 50]
"""
import argparse
import json
import os
import sys
from math import exp

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wild_deep_mvs_amd import synthetic                                      # noqa: E402
from wild_deep_mvs_amd.models.MVSNet.model import build_proj_matrices         # noqa: E402
from wild_deep_mvs_amd.models.trainer import Trainer                          # noqa: E402


def aten_loss(imgs, depth, proj, win):
    b, N, _, h, w = imgs.shape
    inv = torch.inverse(proj)
    ys, xs = torch.meshgrid(torch.arange(h, device=imgs.device), torch.arange(w, device=imgs.device), indexing="ij")
    grid = torch.stack((xs, ys), -1).float().view(1, 1, -1, 2)
    hom = torch.cat((grid, torch.ones_like(grid[..., :1])), -1) * depth.view(b, 1, -1, 1)
    hom = torch.cat((hom, torch.ones_like(hom[..., :1])), -1)
    rp = (hom @ inv[:, 0:1].transpose(2, 3)) @ proj[:, 1:].transpose(2, 3)
    z = rp[..., 2:3]
    fl = (rp[..., :2] / torch.clamp(z, 1e-6)).view(b, N - 1, h, w, 2)
    fl = torch.stack((2 * fl[..., 0] / (w - 1) - 1, 2 * fl[..., 1] / (h - 1) - 1), -1)
    fl = torch.clamp(torch.where((z.view(b, N - 1, h, w, 1) <= 0), torch.full_like(fl, -10.0), fl), -10, 10)
    mask = ((fl < 1).all(-1) & (fl > -1).all(-1)).float()
    conv = lambda x: F.conv2d(x, win, padding=5, groups=3)
    out = []
    a = imgs[:, 0]
    mu1, e11 = conv(a), conv(a * a)
    for i in range(1, N):
        wv = F.grid_sample(imgs[:, i], fl[:, i - 1], align_corners=False)
        mu2 = conv(wv)
        s1, s2, s12 = e11 - mu1 * mu1, conv(wv * wv) - mu2 * mu2, conv(a * wv) - mu1 * mu2
        out.append((1 - ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))).mean(1))
    return torch.stack(out, 1), mask


def timed(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=5)
    ap.add_argument("--height", type=int, default=128)
    ap.add_argument("--width", type=int, default=160)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    sc = synthetic.make_photo_case(a.batch, a.views, a.height, a.width, seed=0)
    imgs = sc["imgs"].cuda()
    proj = build_proj_matrices(sc["K"], sc["R"], sc["t"]).cuda()
    depth0 = sc["depths"][0].cuda()
    g = torch.tensor([exp(-(x - 5) ** 2 / 4.5) for x in range(11)])
    g = (g / g.sum()).unsqueeze(1)
    win = g.mm(g.t()).expand(3, 1, 11, 11).contiguous().cuda()
    tr = Trainer()

    def run(loss_fn):
        d = depth0.clone().requires_grad_(True)
        ssim, mask = loss_fn(d)
        (torch.sum(ssim * mask) / torch.sum(mask)).backward()
        return d.grad

    g_p = run(lambda d: tr.photometricloss(imgs, d, proj))
    g_a = run(lambda d: aten_loss(imgs, d, proj, win))
    rel = ((g_p - g_a).abs().sum() / g_a.abs().sum()).item()
    t_p = timed(lambda: run(lambda d: tr.photometricloss(imgs, d, proj)), a.iters)
    t_a = timed(lambda: run(lambda d: aten_loss(imgs, d, proj, win)), a.iters)
    print(json.dumps({"workload": f"photometric loss fwd+bwd, {a.batch} x {a.views} views, {a.height}x{a.width}",
                      "pscv_ms": round(t_p, 4), "aten_ms": round(t_a, 4), "speedup": round(t_a / t_p, 2),
                      "grad_depth_rel_l1_vs_aten": rel}))


if __name__ == "__main__":
    main()
