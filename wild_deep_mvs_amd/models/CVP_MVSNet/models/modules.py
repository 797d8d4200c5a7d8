"""Drop-in for the reference's ``models/CVP_MVSNet/models/modules.py`` on the pscv engine: intrinsics conditioning and
hypothesis generation (camera / scalar algebra, stays tensor math), ``homo_warping`` / ``proj_cost`` on the fused
HIP warp kernel, and the 3-D block holders of the CVP regulariser."""
from __future__ import annotations

import torch
import torch.nn as nn

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
    hw = tuple(src_feature.shape[2:]) if ref_shape is None else tuple(int(s) for s in ref_shape)
    if src_feature.requires_grad and torch.is_grad_enabled():
        from .... import training as T
        return T.WarpOnlyFn.apply(_cams(ref_in.detach(), [src_in.detach()], ref_ex.detach(), [src_ex.detach()]),
                                  depth_hypos.detach().to(torch.float32).contiguous(), L.GEOM_PROJ, hw, src_feature)
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


def hypo_cams(ref_intrinsics, src_intrinsics, ref_extrinsics, src_extrinsics):
    """fp64 [B,39] camera constants of ``pscv_cvp_depth_hypos`` for ONE source view (tensor-level form; the model's forward gets
    the same block for every level from ``ops.cvp_cams``): K_ref^-1, rows 0..2 of E_src E_ref^-1, K_src, (K_ref R_ref)(K_src R_src)^-1."""
    Ki, Ks = ref_intrinsics.double(), src_intrinsics.double()
    Ei, Es = ref_extrinsics.double(), src_extrinsics.double()
    # closed-form fp64 inverses (adjugate of the 3x3 blocks; extrinsics have the last row (0,0,0,1)): no LAPACK call, so the
    # whole forward stays capturable in a hipGraph
    Ri_inv = ops.inv3x3(Ei[:, :3, :3])
    Ei_inv = torch.zeros_like(Ei)
    Ei_inv[:, :3, :3], Ei_inv[:, :3, 3:4], Ei_inv[:, 3, 3] = Ri_inv, -(Ri_inv @ Ei[:, :3, 3:4]), 1.0
    to_src = Es @ Ei_inv                                                                     # [B,4,4]
    A = (Ki @ Ei[:, :3, :3]) @ ops.inv3x3(Ks @ Es[:, :3, :3])
    return torch.cat((ops.inv3x3(Ki).reshape(-1, 9), to_src[:, :3, :].reshape(-1, 12), Ks.reshape(-1, 9), A.reshape(-1, 9)),
                     dim=1).contiguous()


def calDepthHypo(ref_depths, ref_intrinsics, src_intrinsics, ref_extrinsics, src_extrinsics, depth_min, depth_max, level):
    """Eval-mode hypothesis maps (modules.py:131-226): per batch item, the MEDIAN over valid pixels of the depth step that
    moves the projection into the first source view by one pixel along the epipolar line; 8 planes ``depth + k * step``,
    k = -4..3.  The reference loops over the batch in Python with fp64 tensors; here the per-pixel steps, the exact median
    (radix select) and the planes are three HIP launches with no host round trip (``pscv_cvp_depth_hypos``, scope row
    f-4); only the four 3x3 / 3x4 camera products per batch item are tensor math (fp64 like the reference).
    ref_depths [B,H,W]; src_intrinsics [B,N,3,3]; src_extrinsics [B,N,4,4] -> [B,8,H,W] fp32."""
    cams = hypo_cams(ref_intrinsics, src_intrinsics[:, 0], ref_extrinsics, src_extrinsics[:, 0])
    fallback = ((depth_max - depth_min) / 128).to(torch.float32).reshape(-1).contiguous()
    return ops.cvp_depth_hypos(ref_depths.to(torch.float32).contiguous(), cams, fallback)


def depth_regression(p, depth_values):
    """sum_d p_d depth_d with per-batch planes (modules.py:356-359); tensor-level helper for direct callers."""
    return torch.sum(p * depth_values.view(*depth_values.shape, 1, 1), 1)


def depth_regression_refine(prob_volume, depth_hypothesis):
    """sum_d p_d depth_d with per-pixel planes (modules.py:362-365)."""
    return torch.sum(prob_volume * depth_hypothesis, 1)
