"""Drop-in for the reference's ``models/CVP_MVSNet/models/modules.py`` on the pscv engine: intrinsics conditioning and
hypothesis generation (camera / scalar algebra, stays tensor math), ``homo_warping`` / ``proj_cost`` on the fused
HIP warp kernel, and the 3-D block holders of the CVP regulariser."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import _lib as L
from .... import ops
from ...MVSNet.module import ConvBnReLU3D, ConvBn3D  # noqa: F401  same holders / key names (modules.py:317-337)


def conv(in_planes, out_planes, kernel_size=3, stride=1, padding=1, dilation=1):
    """2-D conv + LeakyReLU(0.1) of the (upstream) feature pyramid (modules.py:24-28)."""
    return nn.Sequential(nn.Conv2d(in_planes, out_planes, kernel_size=kernel_size, stride=stride, padding=padding,
                                   dilation=dilation, bias=True), nn.LeakyReLU(0.1))


def conditionIntrinsics(intrinsics, img_shape, fp_shapes):
    """Rows 0-1 of K divided by image_height / level_height: [B,3,3] -> [B,nScale,3,3] (modules.py:31-50)."""
    levels = []
    for shp in fp_shapes:
        k = intrinsics.clone()
        k[:, :2, :] = k[:, :2, :] / (img_shape[2] / shp[2])
        levels.append(k)
    return torch.stack(levels).permute(1, 0, 2, 3)


def calSweepingDepthHypo(ref_in, src_in, ref_ex, src_ex, depth_min, depth_max, nhypothesis_init=48):
    """``d_i = min + i (max - min) / N`` for i < N (modules.py:53-71)."""
    if nhypothesis_init % 2:
        raise AssertionError("number of depth hypotheses must be even")
    step = (depth_max - depth_min) / nhypothesis_init
    return depth_min.unsqueeze(1) + torch.arange(nhypothesis_init, device=depth_max.device) * step.unsqueeze(1)


def _projection(K, E):
    """[[K E[:3]], [0 0 0 1]] (modules.py:89-93)."""
    top = torch.matmul(K, E[:, 0:3, :])
    last = torch.zeros((K.shape[0], 1, 4), dtype=top.dtype, device=top.device)
    last[:, 0, 3] = 1.0
    return torch.cat((top, last), 1)


def _cams(ref_in, src_ins, ref_ex, src_exs):
    proj = torch.stack([_projection(ref_in, ref_ex)] + [_projection(k, e) for k, e in zip(src_ins, src_exs)], dim=1)
    return ops.proj_cams_device(proj.to(torch.float32).contiguous(), 0)


def homo_warping(src_feature, ref_in, src_in, ref_ex, src_ex, depth_hypos, ref_shape=None):
    """Plane-sweep warp of one source feature map (modules.py:74-128) -> [B,C,D,h,w] fp32; depth_hypos [B,D] or
    [B,D,h,w]."""
    if src_feature.requires_grad and torch.is_grad_enabled():
        raise NotImplementedError("pscv homo_warping: backward is not implemented yet")
    hw = tuple(src_feature.shape[2:]) if ref_shape is None else tuple(int(s) for s in ref_shape)
    fea = ops.to_channels_last(src_feature.detach(), torch.float32)
    vol = ops.warp_cost(None, [fea], _cams(ref_in, [src_in], ref_ex, [src_ex]), depth_hypos.to(torch.float32).contiguous(),
                        geom=L.GEOM_PROJ, cost=L.COST_WARP_ONLY, ref_hw=hw, out_dtype=torch.float32)
    return ops.to_channels_first(vol[0])


def proj_cost(nsrc, ref_feature, src_feature, level, ref_in, src_in, ref_ex, src_ex, depth_hypos,
              storage_dtype=torch.float16, channels_last=False):
    """Refinement cost volume with per-pixel hypotheses (modules.py:229-293) in ONE fused launch.
    ref_feature [B,16,h,w]; src_feature[src][level] [B,16,h,w]; depth_hypos [B,D,h,w]
    -> channels-last variance volume [B,D,h,w,16] (``sum f^2/N - (sum f/N)^2``).
    ``channels_last``: the features already are the engine's [B,h,w,16] 16-bit maps (HIP pyramid tower)."""
    cl = (lambda f: f.contiguous()) if channels_last else (lambda f: ops.to_channels_last(f, storage_dtype))
    srcs = [cl(src_feature[s][level]) for s in range(nsrc)]
    cams = _cams(ref_in, [src_in[:, s] for s in range(nsrc)], ref_ex, [src_ex[:, s] for s in range(nsrc)])
    return ops.warp_cost(cl(ref_feature), srcs, cams,
                         depth_hypos.to(torch.float32).contiguous(), geom=L.GEOM_PROJ, cost=L.COST_VARIANCE_CVP,
                         out_dtype=storage_dtype)


def calDepthHypo(ref_depths, ref_intrinsics, src_intrinsics, ref_extrinsics, src_extrinsics, depth_min, depth_max, level):
    """Eval-mode hypothesis maps (modules.py:131-226): per batch item, the MEDIAN over valid pixels of the depth step
    that moves the projection into the first source view by one pixel along the epipolar line; 8 planes
    ``depth + k * step``, k = -4..3.  fp64 like the reference; scalar / per-pixel algebra, stays tensor math
    (scope row f-4).  ref_depths [B,H,W]; src_intrinsics [B,N,3,3]; src_extrinsics [B,N,4,4] -> [B,8,H,W] fp32."""
    B, H, W = ref_depths.shape
    dev = ref_depths.device
    Ki, Ks = ref_intrinsics.double(), src_intrinsics[:, 0].double()
    Ei, Es = ref_extrinsics.double(), src_extrinsics[:, 0].double()
    # pixel order x-major (the reference builds meshgrid(x, y) and transposes the depth map to match)
    xs = torch.arange(W, device=dev, dtype=torch.float64).repeat_interleave(H)
    ys = torch.arange(H, device=dev, dtype=torch.float64).repeat(W)
    X = torch.stack([xs, ys, torch.ones_like(xs)], 0)                                   # [3,HW]
    out = ref_depths.unsqueeze(1).repeat(1, 8, 1, 1)
    for b in range(B):
        d1 = ref_depths[b].transpose(0, 1).reshape(-1).double()
        to_src = Es[b] @ torch.linalg.inv(Ei[b])
        Kinv = torch.linalg.inv(Ki[b])

        def project(depth):
            cam = Kinv @ (X * depth)
            p = Ks[b] @ (to_src[:3, :3] @ cam + to_src[:3, 3:4])
            return p / p[2:3], p[2]

        x1, z1 = project(d1)
        x2, z2 = project(d1 + 1)
        direction = x2 - x1
        norm = torch.linalg.norm(direction, dim=0)
        x3 = x1 + direction / norm.clamp(min=1e-8)
        A = (Ki[b] @ Ei[b][:3, :3]) @ torch.linalg.inv(Ks[b] @ Es[b][:3, :3])
        rhs, col2 = z1 * (A @ x1), A @ x3
        # rows 1..2 of [X | A x3] * (delta, .)^T = rows 1..2 of z1 A x1, solved with Cramer's rule
        m00, m01, m10, m11 = X[1], col2[1], X[2], col2[2]
        det = m00 * m11 - m01 * m10
        valid = (norm > 1e-8) & (z1 > 1e-8) & (z2 > 1e-8) & (det.abs() > 1e-8)
        if bool(valid.any()):
            delta = ((m11 * rhs[1] - m01 * rhs[2]) / det)[valid]
            step = delta.abs().median()
        else:
            step = ((depth_max - depth_min) / 128).double().reshape(-1)[0]
        for k in range(-4, 4):
            out[b, k + 4] = (ref_depths[b].double() + k * step).to(out.dtype)
    return out.float()


def depth_regression(p, depth_values):
    """sum_d p_d depth_d with per-batch planes (modules.py:356-359); tensor-level helper for direct callers."""
    return torch.sum(p * depth_values.view(*depth_values.shape, 1, 1), 1)


def depth_regression_refine(prob_volume, depth_hypothesis):
    """sum_d p_d depth_d with per-pixel planes (modules.py:362-365)."""
    return torch.sum(prob_volume * depth_hypothesis, 1)
