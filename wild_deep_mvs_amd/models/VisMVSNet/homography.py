"""Drop-in for the reference's ``models/VisMVSNet/homography.py`` function-level API on the pscv engine.

Inside the model the per-plane 3x3 homographies are never materialised: ``pscv_homog_cams`` reduces the cameras to
two 3x3 blocks (A, Bm) per (source, batch) and the warp kernel evaluates ``A p - Bm p / (d + 1e-9)`` per voxel.
``get_homographies`` / ``homography_warping`` are kept for callers that use them directly."""
from __future__ import annotations

import torch

from ... import _lib as L
from ... import ops


def get_homographies(left_cam, right_cam, depth_num, depth_start, depth_interval, inv=False):
    """[n,d,1|h,1|w,3,3] plane-induced homographies (reference homography.py:23-74).  Camera algebra (row A0 of
    the scope table): plain tensor math on the caller's device."""
    n = left_cam.shape[0]
    R_l, R_r = left_cam[:, 0, :3, :3], right_cam[:, 0, :3, :3]
    t_l, t_r = left_cam[:, 0, :3, 3:4], right_cam[:, 0, :3, 3:4]
    K_l, K_r = left_cam[:, 1, :3, :3], right_cam[:, 1, :3, :3]
    steps = torch.arange(depth_num, dtype=left_cam.dtype, device=left_cam.device).view(1, depth_num, 1, 1)
    if not inv:
        depth = depth_start + depth_interval * steps
    else:                                   # planes uniform in INVERSE depth between the same end points (homography.py:41-46)
        depth_end = depth_start + (depth_num - 1) * depth_interval
        inv_interv = (1 / (depth_start + 1e-9) - 1 / (depth_end + 1e-9)) / (depth_num - 1 + 1e-9)
        depth = 1 / (1 / (depth_end + 1e-9) + inv_interv * steps)
    depth = depth[..., None, None]
    c_rel = (-R_r.transpose(-2, -1) @ t_r) - (-R_l.transpose(-2, -1) @ t_l)
    plane = (c_rel @ R_l[:, 2:3, :3]).view(n, 1, 1, 1, 3, 3)
    eye = torch.eye(3, dtype=left_cam.dtype, device=left_cam.device).view(1, 1, 1, 1, 3, 3)
    back = (R_l.transpose(-2, -1) @ ops.inv3x3(K_l.double()).to(left_cam.dtype)).view(n, 1, 1, 1, 3, 3)
    H = (K_r @ R_r).view(n, 1, 1, 1, 3, 3) @ ((eye - plane / (depth + 1e-9)) @ back)
    if torch.isnan(H).any():
        raise Exception("Nan")
    return H


def homography_warping(input, H, ref_shape=None):
    """Warp ``input`` [m,c,hs,ws] with H [m,3,3] / [m,1,1,3,3] (one homography per batch item) or [m,h,w,3,3] (one per
    reference pixel) -- the shapes of homography.py:107-120 -- -> [m,c,h,w] fp32.  Half-pixel centres, ``z <= 0`` -> zero sample,
    index ``u (W-1)/W``.  Inside the model per-pixel homographies never exist (``SingleStage`` hands per-pixel depth planes to the
    fused sweep); this function-level form runs the small ``pscv_homography_warp`` kernel (``pscv_homography_warp_bwd`` under autograd)."""
    if H.dim() == 5 and not (H.shape[1] == 1 and H.shape[2] == 1):
        hw = tuple(input.shape[2:]) if ref_shape is None else tuple(int(s) for s in ref_shape)
        if tuple(H.shape[1:3]) != hw:
            raise ValueError(f"pscv homography_warping: per-pixel H {tuple(H.shape)} does not match the reference shape {hw}")
        if input.requires_grad and torch.is_grad_enabled():     # gradient to `input` only, like grid_sample under the reference's no_grad grid
            from ... import training as T
            return T.HomographyWarpFn.apply(H, hw, input)
        out = ops.homography_warp(ops.to_channels_last(input.detach(), torch.float32), H.detach().to(torch.float32).contiguous(), hw)
        return out.permute(0, 3, 1, 2)
    if H.dim() == 5:
        H = H.view(-1, 3, 3)
    m = input.shape[0]
    hw = tuple(input.shape[2:]) if ref_shape is None else tuple(int(s) for s in ref_shape)
    cams = torch.zeros((1, m, L.CAM_FLOATS), dtype=torch.float32, device=input.device)
    cams[0, :, :9] = H.reshape(m, 9).to(torch.float32)            # hom = A p - 0 / (d + 1e-9) with d = 1
    one = torch.ones((m, 1), dtype=torch.float32, device=input.device)
    if input.requires_grad and torch.is_grad_enabled():
        from ... import training as T
        return T.WarpOnlyFn.apply(cams, one, L.GEOM_HOMOG, hw, input)[:, :, 0]
    fea = ops.to_channels_last(input.detach(), torch.float32)
    vol = ops.warp_cost(None, [fea], cams, one, geom=L.GEOM_HOMOG, cost=L.COST_WARP_ONLY, ref_hw=hw,
                        out_dtype=torch.float32)                   # [1,m,1,h,w,c]
    return vol[0, :, 0].permute(0, 3, 1, 2)
