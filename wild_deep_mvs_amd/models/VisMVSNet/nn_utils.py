"""Drop-in for the live parts of the reference's ``models/VisMVSNet/nn_utils.py``: the block / U-Net containers
(same state-dict keys as nn_utils.py:123-278) and the function-level ``soft_argmin`` / ``entropy`` /
``groupwise_correlation`` on the pscv engine.  2-D instances (feature extractor) run on PyTorch-ROCm; 3-D instances
are parameter holders executed by ``model_cas.Reg*`` through MFMA conv launches."""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

from ... import ops


class BasicBlock(nn.Module):
    """conv3-bn-relu-conv3-bn + (strided 1x1 conv-bn | identity), relu.  Keys: conv1, bn1, conv2, bn2, downsample.{0,1}."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, dim=2):
        super().__init__()
        conv = nn.Conv2d if dim == 2 else nn.Conv3d
        norm = nn.BatchNorm2d if dim == 2 else nn.BatchNorm3d
        self.conv1 = conv(inplanes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = norm(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = norm(planes)
        self.downsample = downsample
        self.stride, self.dim = stride, dim

    def forward(self, x):   # 2-D only (3-D blocks are executed by the engine)
        if self.dim != 2:
            raise RuntimeError("3-D blocks are executed by the pscv engine")
        y = self.bn2(self.conv2(self.relu(self.bn1(self.conv1(x)))))
        return self.relu(y + (x if self.downsample is None else self.downsample(x)))


def _make_layer(inplanes, block, planes, blocks, stride=1, dim=2):
    conv = nn.Conv2d if dim == 2 else nn.Conv3d
    norm = nn.BatchNorm2d if dim == 2 else nn.BatchNorm3d
    shortcut = None
    if stride != 1 or inplanes != planes * block.expansion:
        shortcut = nn.Sequential(conv(inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                                 norm(planes * block.expansion))
    seq = [block(inplanes, planes, stride, shortcut, dim=dim)]
    seq += [block(planes * block.expansion, planes, dim=dim) for _ in range(1, blocks)]
    return nn.Sequential(*seq)


class UNet(nn.Module):
    """Encoder (first level stride 1, then stride 2) / decoder (deconv -> cat([deconv, skip]) -> conv [-> blocks]).
    Registered names follow the reference (``{prefix}{scale}_{idx}`` under enc_blocks / dec_blocks)."""

    def __init__(self, inplanes: int, enc: int, dec: int, initial_scale: int, bottom_filters: List[int],
                 filters: List[int], head_filters: List[int], prefix: str, dim: int = 2):
        super().__init__()
        if bottom_filters or head_filters:
            raise NotImplementedError("bottom / head blocks are unused by the reference's models")
        conv = nn.Conv2d if dim == 2 else nn.Conv3d
        deconv = nn.ConvTranspose2d if dim == 2 else nn.ConvTranspose3d
        self.dim, self.dec = dim, dec
        scale, idx, prev = initial_scale, 0, inplanes
        self.enc_blocks = nn.ModuleDict()
        for f in filters:
            self.enc_blocks[f"{prefix}{scale}_{idx}"] = _make_layer(prev, BasicBlock, f, enc, 1 if idx == 0 else 2, dim=dim)
            idx, scale, prev = idx + 1, scale * 2, f
        self.dec_blocks = nn.ModuleDict()
        for f in filters[-2::-1]:
            parts = [deconv(prev, f, 3, 2, 1, 1, bias=False), conv(2 * f, f, 3, 1, 1, bias=False)]
            if dec > 0:
                parts.append(_make_layer(f, BasicBlock, f, dec, 1, dim=dim))
            self.dec_blocks[f"{prefix}{scale}_{idx}"] = nn.ModuleList(parts)
            idx, scale, prev = idx + 1, scale // 2, f

    def forward(self, x, multi_scale=1):   # 2-D only
        if self.dim != 2:
            raise RuntimeError("3-D U-Nets are executed by the pscv engine")
        skips = []
        for blk in self.enc_blocks.values():
            x = blk(x)
            skips.append(x)
        outs = [x]
        for i, parts in enumerate(self.dec_blocks.values()):
            x = parts[1](torch.cat([parts[0](x), skips[-2 - i]], 1))
            if len(parts) == 3:
                x = parts[2](x)
            outs.append(x)
        return x if multi_scale == 1 else outs[-multi_scale:]


# ---- function-level API (reference nn_utils.py:453-490), dim = 1 on [n,d,h,w] volumes ------------------
def soft_argmin(volume, dim, keepdim=False, window=None):
    """Returns (prob_volume, expected index[, window probability]) like the reference (nn_utils.py:453-466)."""
    if dim != 1 or volume.dim() != 4:
        raise NotImplementedError("pscv soft_argmin: [n,d,h,w] volumes with dim=1")
    o = ops.softargmin(volume.contiguous(), None, want_index=True, want_prob=True, want_conf=window is not None,
                       conf_mode=1, window=float(window or 0))
    idx = o["index"].unsqueeze(1) if keepdim else o["index"]
    if window is None:
        return o["prob"], idx
    return o["prob"], idx, (o["conf"].unsqueeze(1) if keepdim else o["conf"])


def entropy(volume, dim, keepdim=False):
    """-sum p log(clamp(p, 1e-9, 1)) of a probability volume (nn_utils.py:469-470).  Elementwise helper kept for
    direct callers; the model takes the entropy from the fused softargmin kernel."""
    return torch.sum(-volume * volume.clamp(1e-9, 1.).log(), dim=dim, keepdim=keepdim)


def groupwise_correlation(v1, v2, groups, dim):
    """Sum of channel products inside each group (nn_utils.py:473-490).  Elementwise helper kept for direct callers;
    the model gets this volume straight out of the fused warp kernel (PSCV_COST_GROUPCORR)."""
    shape = list(v1.shape)
    c = shape[dim]
    if c % groups:
        raise AssertionError("channels must divide into groups")
    new = shape[:dim] + [groups, c // groups] + shape[dim + 1:]
    return (v1.reshape(new) * v2.reshape(new)).sum(dim=dim + 1)
