"""The helpers of the reference's ``utils/utils_3D.py`` that its ``models/trainer.py`` pulls into its namespace
(``build_grid``, ``build_proj_matrices``, ``normalize``, ``flows_from_single_depthmap``: utils_3D.py:29-62,185-208,243-273).
Function-level torch forms for callers that import them from ``models.trainer``; inside the loss the same geometry runs fused
in ``pscv_photo_warp`` (csrc/photo_loss.hip) and the flows are never materialised."""
from __future__ import annotations

import numpy as np
import torch

from ..models.MVSNet.model import build_proj_matrices  # noqa: F401  (utils_3D.py:50-62)


def build_grid(h, w, device, normed=True):
    """[1,h,w,2] grid of (x, y): pixel indices, or coordinates in [-1, 1] when ``normed``; numpy when ``device`` is None
    (utils_3D.py:29-47)."""
    if device is None:
        ys = np.linspace(-1, 1, h) if normed else np.arange(h)
        xs = np.linspace(-1, 1, w) if normed else np.arange(w)
        gx, gy = np.meshgrid(xs, ys)
        return np.stack((gx, gy), axis=-1)[None]
    ys = torch.linspace(-1, 1, steps=h, device=device) if normed else torch.arange(h, device=device)
    xs = torch.linspace(-1, 1, steps=w, device=device) if normed else torch.arange(w, device=device)
    return torch.stack((xs.view(1, w).expand(h, w), ys.view(h, 1).expand(h, w)), dim=-1).unsqueeze(0)


def normalize(flow, h, w, clamp=None):
    """Pixel coordinates -> ``2 x / (size - 1) - 1`` on the last axis ((x, y) or (x, y, 1)), optionally clamped to +-clamp;
    ``h`` / ``w`` are numbers or per-batch tensors (utils_3D.py:243-273)."""
    if not torch.is_tensor(h):
        h = torch.tensor(float(h), device=flow.device).view(1)
        w = torch.tensor(float(w), device=flow.device).view(1)
    lead = {3: (-1, 1), 4: (-1, 1, 1), 5: (1, -1, 1, 1)}.get(flow.dim())
    if lead is not None:
        h, w = h.reshape(lead), w.reshape(lead)
    out = torch.empty_like(flow)
    if out.shape[-1] == 3:
        out[..., 2] = 1
    out[..., 0] = 2 * flow[..., 0] / (w - 1) - 1
    out[..., 1] = 2 * flow[..., 1] / (h - 1) - 1
    return torch.clamp(out, -clamp, clamp) if clamp else out


def flows_from_single_depthmap(depthmaps, proj_mat, ref_idx):
    """depthmaps [b,h,w] of view ``ref_idx``, proj_mat [b,N,4,4] -> pixel positions of every reference pixel in the other
    N-1 views [b,N-1,h,w,2] and their depth there [b,N-1,h,w] (the divisor clamped at 1e-6)  (utils_3D.py:185-208)."""
    b, N = proj_mat.shape[:2]
    _, h, w = depthmaps.shape
    dev = proj_mat.device
    src = [i for i in range(N) if i != ref_idx]
    ys, xs = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing="ij")
    pix = torch.stack((xs, ys, torch.ones_like(xs)), dim=-1).float().view(1, 1, h * w, 3)
    cam = torch.cat((pix * depthmaps.reshape(b, 1, h * w, 1), torch.ones(b, 1, h * w, 1, device=dev)), dim=-1)
    world = cam @ torch.inverse(proj_mat)[:, ref_idx:ref_idx + 1].transpose(2, 3)
    proj = world @ proj_mat[:, src].transpose(2, 3)
    z = proj[..., 2:3]
    return (proj[..., :2] / torch.clamp(z, 1e-6)).view(b, N - 1, h, w, 2), z.view(b, N - 1, h, w)
