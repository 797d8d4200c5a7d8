"""Drop-in for the pieces of the reference's ``models/utils.py`` that touch the plane-sweep path.

``homo_warp`` is the function-level name BASELINE.json's north star gives the differentiable plane-sweep warp; the reference
itself only has ``models.MVSNet.module.homo_warping``, so the alias points there.  ``rec_upsample`` and
``bayesian_version_loss`` are the two helpers the loss code of ``models/trainer.py`` takes from this module
(models/utils.py:101-119): plain tensor plumbing around the engine's outputs, kept in torch."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .MVSNet.module import homo_warping

homo_warp = homo_warping


def rec_upsample(vars, size):
    """Bilinear upsampling applied recursively to lists / tuples / dicts of [b,h,w] or [b,c,h,w] maps; ``None`` passes through
    (models/utils.py:46-58, 101-107)."""
    if isinstance(vars, list):
        return [rec_upsample(v, size) for v in vars]
    if isinstance(vars, tuple):
        return tuple(rec_upsample(v, size) for v in vars)
    if isinstance(vars, dict):
        return {k: rec_upsample(v, size) for k, v in vars.items()}
    if vars is None:
        return None
    if vars.dim() == 3:
        return F.interpolate(vars.unsqueeze(1), size=size, mode="bilinear", align_corners=False).squeeze(1)
    return F.interpolate(vars, size=size, mode="bilinear", align_corners=False)


def bayesian_version_loss(l, u, mask):
    """``sum((l exp(-u) + u) mask) / sum(mask) + sum(l mask) / sum(mask)``; un-normalised when the mask is empty, which keeps
    the graph alive (models/utils.py:110-119)."""
    mask_sum = torch.sum(mask)
    uncert_loss = torch.sum((l * torch.exp(-u) + u) * mask)
    org_loss = torch.sum(l * mask)
    if mask_sum != 0:
        return uncert_loss / mask_sum + org_loss / mask_sum
    return uncert_loss + org_loss
