"""Drop-in for the reference's ``models/MVSNet/module.py``: the 3-D block containers (parameter holders
with the reference's state-dict names) and the function-level hot-path API ``homo_warping`` /
``depth_regression``, executed by the pscv HIP engine."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import _lib as L
from ... import ops


class ConvBnReLU(nn.Module):
    """2-D conv + BN + ReLU of the (upstream) feature extractor; plain PyTorch-ROCm / MIOpen.
    Same parameter names as reference models/MVSNet/module.py:21-28 (``conv``, ``bn``)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, pad=1):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=pad, bias=False)
        self.bn = nn.BatchNorm2d(out_channels)

    def forward(self, x):
        return F.relu(self.bn(self.conv(x)), inplace=True)


class _Block3D(nn.Module):
    """Parameter holder for one 3x3x3 conv + BatchNorm3d.  The engine reads ``conv.weight`` and the BN
    statistics (folded into the conv epilogue); ``forward`` is never routed through ATen."""
    relu = True

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, pad=1):
        super().__init__()
        if kernel_size != 3 or pad != 1 or stride not in (1, 2):
            raise ValueError("pscv conv3d supports 3x3x3, padding 1, stride 1 or 2")
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size, stride=stride, padding=pad, bias=False)
        self.bn = nn.BatchNorm3d(out_channels)
        self.stride = stride

    def engine_layer(self, device, dtype=torch.float16) -> ops.Conv3dLayer:
        bn = self.bn
        return ops.Conv3dLayer.build(self.conv.weight, kind=L.CONV_S1 if self.stride == 1 else L.CONV_S2,
                                     device=device, bn=(bn.weight, bn.bias, bn.running_mean, bn.running_var),
                                     bn_eps=bn.eps, relu=self.relu, dtype=dtype)

    def forward(self, x):
        raise RuntimeError("3-D blocks are executed by the pscv engine (CostRegNet.forward), not called directly")


class ConvBnReLU3D(_Block3D):   # reference models/MVSNet/module.py:41-48
    relu = True


class ConvBn3D(_Block3D):       # reference models/MVSNet/module.py:51-58
    relu = False


def deconv_engine_layer(seq: nn.Sequential, device, stride: int = 2, dtype=torch.float16) -> ops.Conv3dLayer:
    """``Sequential(ConvTranspose3d(k3,p1,op=stride-1), BatchNorm3d, ReLU)`` -> engine layer
    (reference models/MVSNet/model.py:57-70)."""
    deconv, bn = seq[0], seq[1]
    return ops.Conv3dLayer.build(deconv.weight, kind=L.CONV_T2 if stride == 2 else L.CONV_S1, transposed=True,
                                 device=device, bn=(bn.weight, bn.bias, bn.running_mean, bn.running_var),
                                 bn_eps=bn.eps, relu=True, dtype=dtype)


def homo_warping(src_fea, src_proj, ref_proj, depth_values, ref_shape=None):
    """Plane-sweep warp of one source feature map (reference models/MVSNet/module.py:111-169).

    src_fea [B,C,Hs,Ws]; src_proj, ref_proj [B,4,4]; depth_values [B,D] or [B,D,h,w]
    -> [B,C,D,h,w] fp32 (an NCDHW-indexed view of the engine's channels-last volume).
    Differentiable w.r.t. ``src_fea`` (the grid is built under no_grad in the reference too)."""
    cams = ops.proj_cams_device(torch.stack([ref_proj, src_proj], dim=1).detach().to(torch.float32).contiguous(), 0)
    hw = src_fea.shape[-2:] if ref_shape is None else tuple(int(s) for s in ref_shape)
    if src_fea.requires_grad and torch.is_grad_enabled():
        from ... import training as T
        return T.WarpOnlyFn.apply(cams, depth_values.detach().to(torch.float32).contiguous(), L.GEOM_PROJ, tuple(hw), src_fea)
    fea = ops.to_channels_last(src_fea.detach(), torch.float32)
    vol = ops.warp_cost(None, [fea], cams, depth_values.to(torch.float32).contiguous(), geom=L.GEOM_PROJ,
                        cost=L.COST_WARP_ONLY, ref_hw=hw, out_dtype=torch.float32)
    return ops.to_channels_first(vol[0])


def depth_regression(p, depth_values):
    """sum_d p_d * depth_d over dim 1 (reference models/MVSNet/module.py:174-178).  Kept as the
    reference's tensor-level helper for callers that already hold a probability volume; the model's own
    forward uses the fused ``ops.softargmin``."""
    if depth_values.dim() <= 3:
        depth_values = depth_values.view(*depth_values.shape, 1, 1)
    return torch.sum(p * depth_values, 1)
