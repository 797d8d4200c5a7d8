"""Drop-in for the reference's ``models/utils.py``: every public name of that module exists here with the same call
signature and result (``from models.utils import *`` in depthmap_eval.py / evaluation/run_depthmaps.py / models/trainer.py keeps
working when this package is installed as ``models``).

``homo_warp`` is the function-level name BASELINE.json's north star gives the differentiable plane-sweep warp; the reference
itself only has ``models.MVSNet.module.homo_warping``, so the alias points there -- that one runs on the
HIP engine.  Everything else is host-side plumbing around the engine's outputs (nested-container mapping, device moves,
evaluation metrics) and stays in torch, written for this package (one generic container mapper instead of the reference's
decorator pair; the decorator names are kept because ``import *`` exports them)."""
from __future__ import annotations

import numpy as np
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


# ---- nested containers (models/utils.py:23-99) ------------------------------------------------------------------------------
def _map_nested(fn, obj, *extra):
    """Applies ``fn(leaf, *extra)`` to every leaf of nested lists / tuples / dicts, preserving the container types."""
    if isinstance(obj, list):
        return [_map_nested(fn, x, *extra) for x in obj]
    if isinstance(obj, tuple):
        return tuple(_map_nested(fn, x, *extra) for x in obj)
    if isinstance(obj, dict):
        return {k: _map_nested(fn, v, *extra) for k, v in obj.items()}
    return fn(obj, *extra)


def make_recursive_func(func):
    """``func(leaf)`` -> function over nested lists / tuples / dicts (models/utils.py:32-43)."""
    return lambda vars: _map_nested(func, vars)


def make_recursive_func2(func):
    """``func(leaf, param)`` -> function over nested containers with one shared parameter (models/utils.py:46-58)."""
    return lambda vars, param: _map_nested(func, vars, param)


def make_nograd_func(func):
    """Runs ``func`` under ``torch.no_grad()`` (models/utils.py:23-29)."""
    def wrapper(*args, **kwargs):
        with torch.no_grad():
            return func(*args, **kwargs)
    return wrapper


def _leaf_error(name, x):
    return NotImplementedError(f"invalid input type {type(x)} for {name}")


def _to_float(x):
    if isinstance(x, float):
        return x
    if isinstance(x, torch.Tensor):
        return x.data.item()
    raise _leaf_error("tensor2float", x)


def _to_numpy(x):
    if isinstance(x, np.ndarray):
        return x
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy().copy()
    raise _leaf_error("tensor2numpy", x)


def _to_cuda(x):
    if isinstance(x, torch.Tensor):
        return x.cuda()
    if isinstance(x, str):
        return x
    raise _leaf_error("tocuda", x)


def _add_batch(x):
    if isinstance(x, torch.Tensor):
        return x.unsqueeze(0)
    if isinstance(x, str):
        return x
    raise _leaf_error("add_batch", x)


tensor2float = make_recursive_func(_to_float)        # models/utils.py:61-68
tensor2numpy = make_recursive_func(_to_numpy)        # :71-78
tocuda = make_recursive_func(_to_cuda)               # :81-88  (strings pass through: scene / file names inside a sample)
add_batch = make_recursive_func(_add_batch)          # :91-98


# ---- evaluation metrics (models/utils.py:123-171): computed per image of the batch, then averaged; never part of a loss ----------
def compute_metrics_for_each_image(metric_func):
    """``metric_func(est, gt, mask, *args)`` on single images -> mean over the batch (models/utils.py:123-135)."""
    def wrapper(depth_est, depth_gt, mask, *args):
        per_image = [metric_func(depth_est[i], depth_gt[i], mask[i], *args) for i in range(depth_gt.shape[0])]
        return torch.stack(per_image).mean()
    return wrapper


def _metric(fn):
    return make_nograd_func(compute_metrics_for_each_image(fn))


def _thres(est, gt, mask, thres):
    assert isinstance(thres, (int, float))
    return ((est[mask] - gt[mask]).abs() > thres).float().mean()


def _rel_thres(est, gt, mask, thres):
    assert isinstance(thres, (int, float))
    e, g = est[mask], gt[mask]
    return 1 - (torch.max(e / g, g / e) > thres).float().mean()


Thres_metrics = _metric(_thres)                                                                  # fraction with |est - gt| > thres
Rel_Thres_metrics = _metric(_rel_thres)                                                          # fraction with max(est/gt, gt/est) <= thres
AbsDepthError_metrics = _metric(lambda est, gt, mask: (est[mask] - gt[mask]).abs().mean())
RelDepthError_metrics = _metric(lambda est, gt, mask: ((est[mask] - gt[mask]).abs() / gt[mask]).mean())
SquareRelDepthError_metrics = _metric(lambda est, gt, mask: ((est[mask] - gt[mask]) ** 2 / gt[mask]).mean())
