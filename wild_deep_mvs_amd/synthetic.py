"""Synthetic plane-sweep inputs and deterministic network weights.

There are no datasets or checkpoints in the build environment, so the tests,
``bench.py``, ``__graft_entry__.smoke()`` and the golden-vector generator
(``tests/golden/gen_golden.py``) all draw their cameras, images, feature maps
and weights from here.  Everything is seeded through ``numpy.random.Generator``
(PCG64, platform independent) so that the same tensors can be re-created on the
GPU box without committing them.

The sample dict mirrors what the reference's dataset loaders hand to
``forward()`` (reference ``data/dtu_yao.py:133-146``): ``imgs`` [B,V,3,H,W],
``K``/``R`` [B,V,3,3], ``t`` [B,V,3,1], ``depth_min``/``depth_max`` [B,V].

Weights are *sharpened*: with PyTorch default initialisation the reference's
logits have a std of ~1e-6 over the depth axis, the softmax is uniform and the
depth map is independent of the warp, which would make
every depth-parity check vacuous.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Mapping, Sequence, Tuple

import numpy as np
import torch


# --------------------------------------------------------------------------
# cameras / scenes
# --------------------------------------------------------------------------
def make_cameras(B: int, V: int, H: int, W: int, *, depth_min: float = None,
                 depth_max: float = None, behind_view: int = -1, rig: str = "probe",
                 dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Two pinhole rigs.

    ``rig="probe"`` (default; depth 2..6): reference camera at identity, source ``v`` rotated about the
    y axis by ``0.05 v (-1)^v`` rad and shifted ``0.1 v (-1)^v`` along x.

    ``rig="dtu"`` (depth 425..905 = 425 + 192 x 2.5, reference ``data/dtu_yao.py:109``): the geometry of
    Yao's DTU training set -- focal length 2.26 W (1446 px at 640), cameras on an arc around an object
    point 660 mm in front of the reference camera, looking at it: source ``v`` is rotated about the
    vertical axis through that point by ``5 ceil(v/2) (-1)^v`` degrees (baselines 58 / 115 mm) and tilted
    by ``1.5 v (-1)^(v//2)`` degrees about the horizontal axis, so epipolar lines are not image rows and
    one plane step moves a sample by 0.13-0.27 feature texels (0.03-0.1 in the probe rig).

    ``behind_view >= 0`` turns that source camera around (rotation by ~pi about
    y) so that every reference ray lands behind it -- the ``q_z <= 0`` branch of
    the warp (reference ``models/MVSNet/module.py:147-150``).
    """
    if rig not in ("probe", "dtu"):
        raise ValueError(f"unknown rig {rig!r}")
    if depth_min is None:
        depth_min = 2.0 if rig == "probe" else 425.0
    if depth_max is None:
        depth_max = 6.0 if rig == "probe" else 905.0
    K = torch.zeros(B, V, 3, 3, dtype=dtype)
    R = torch.zeros(B, V, 3, 3, dtype=dtype)
    t = torch.zeros(B, V, 3, 1, dtype=dtype)
    for b in range(B):
        for v in range(V):
            sgn = -1.0 if v % 2 else 1.0
            if rig == "probe":
                f = 0.9 * W * (1.0 + 0.02 * v + 0.01 * b)
                K[b, v] = torch.tensor([[f, 0.0, W / 2.0 + 0.5 * v],
                                        [0.0, f, H / 2.0 - 0.25 * v],
                                        [0.0, 0.0, 1.0]], dtype=dtype)
                a = 0.05 * v * sgn + 0.01 * b
                if v == behind_view:
                    a = math.pi - 0.1
                ca, sa = math.cos(a), math.sin(a)
                R[b, v] = torch.tensor([[ca, 0.0, sa], [0.0, 1.0, 0.0], [-sa, 0.0, ca]], dtype=dtype)
                t[b, v] = torch.tensor([[0.1 * v * sgn], [0.02 * v], [0.0]], dtype=dtype)
            else:
                f = 2.2595 * W * (1.0 + 0.002 * v)
                K[b, v] = torch.tensor([[f, 0.0, W / 2.0 + 11.6 * W / 640.0 + 0.5 * v],
                                        [0.0, f, H / 2.0 + 9.5 * H / 512.0 - 0.25 * v],
                                        [0.0, 0.0, 1.0]], dtype=dtype)
                zc = 660.0 + 5.0 * b
                th = math.radians(5.0 * ((v + 1) // 2)) * sgn + 0.002 * b
                ph = math.radians(1.5 * v) * (-1.0 if (v // 2) % 2 else 1.0)
                if v == behind_view:
                    th = math.pi - 0.1
                ct, st_, cp, sp = math.cos(th), math.sin(th), math.cos(ph), math.sin(ph)
                Ry = torch.tensor([[ct, 0.0, st_], [0.0, 1.0, 0.0], [-st_, 0.0, ct]], dtype=torch.float64)
                Rx = torch.tensor([[1.0, 0.0, 0.0], [0.0, cp, -sp], [0.0, sp, cp]], dtype=torch.float64)
                Rv = Rx @ Ry
                pivot = torch.tensor([[0.0], [0.0], [zc]], dtype=torch.float64)
                R[b, v] = Rv.to(dtype)
                t[b, v] = (pivot - Rv @ pivot).to(dtype)          # x_cam = Rv (x - pivot) + pivot: every camera sees the pivot at (0, 0, zc)
    dmin = torch.full((B, V), float(depth_min), dtype=dtype)
    dmax = torch.full((B, V), float(depth_max), dtype=dtype)
    return {"K": K, "R": R, "t": t, "depth_min": dmin, "depth_max": dmax}


def make_filter_scene(V: int, H: int, W: int, *, seed: int = 0, behind_view: int = -1, half_res_view: int = -1,
                      near_view: int = -1, baseline: float = 4.0) -> Dict[str, object]:
    """Depth maps of one tilted world plane seen by the ``make_cameras`` rig, for the geometric-consistency filter
    (the step after the hot path): ``depth`` [H,W] of view 0, ``src_depth`` list of V-1 maps, ``K``, ``R`` [V,3,3],
    ``t`` [V,3,1].  The plane depth is analytic per view (``d = (c + n.R^T t) / (n.R^T K^-1 p)``), so the maps are
    mutually consistent; smooth multiplicative perturbations of 0-3 % (independent per view) then move pixels to
    either side of the filter's 1 % / 1 px thresholds, and a block of gross outliers is planted in source 1.
    ``half_res_view`` renders that source at half resolution (own intrinsics): the filter takes per-source shapes.
    ``baseline`` scales the camera translations (wider baseline = larger triangulation angles); ``near_view`` keeps
    that source almost at the reference's position (triangulation angle below the 1 degree default)."""
    cam = make_cameras(1, V, H, W, behind_view=behind_view)
    K, R, t = cam["K"][0].clone(), cam["R"][0].clone(), cam["t"][0].clone() * baseline
    if near_view >= 0:
        t[near_view] = t[near_view] * 0.02
        R[near_view] = torch.eye(3)
    rng = np.random.default_rng(seed)
    n = torch.tensor([0.10, -0.06, 1.0])
    n = n / n.norm()
    c = 4.0 * float(n[2])
    maps = []
    for v in range(V):
        h, w = H, W
        Kv = K[v].clone()
        if v == half_res_view:
            h, w = H // 2, W // 2
            Kv[:2] *= 0.5
            K[v] = Kv
        ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
        p = torch.stack((xs, ys, torch.ones_like(xs)), dim=-1).reshape(-1, 3)
        ray = (p @ torch.inverse(Kv).t()) @ R[v]                       # R^T K^-1 p, as rows
        num = c + float(n @ (R[v].t() @ t[v]).squeeze(-1))
        d = (num / (ray @ n)).reshape(h, w)
        coarse = torch.from_numpy(rng.standard_normal((1, 1, max(h // 12, 2), max(w // 12, 2))).astype(np.float32))
        bump = torch.nn.functional.interpolate(coarse, size=(h, w), mode="bilinear", align_corners=False)[0, 0]
        d = d * (1.0 + 0.012 * bump * (0.3 if v == 0 else 1.0))
        if v == 1:
            d[h // 4:h // 2, w // 3:w // 2] *= 1.4                    # gross outliers
        maps.append(d.contiguous())
    return {"depth": maps[0], "src_depth": maps[1:], "K": K, "R": R, "t": t}


def make_scene(B: int, V: int, H: int, W: int, *, seed: int = 0, depth_min: float = None,
               depth_max: float = None, behind_view: int = -1, rig: str = "probe") -> Dict[str, torch.Tensor]:
    """Full sample dict with smooth-ish random images in [0, 1)."""
    rng = np.random.default_rng(seed)
    # low-frequency content + noise so that the 2D feature nets see structure
    coarse = rng.random((B, V, 3, max(H // 8, 1), max(W // 8, 1)), dtype=np.float32)
    coarse_t = torch.from_numpy(coarse).reshape(B * V, 3, coarse.shape[-2], coarse.shape[-1])
    smooth = torch.nn.functional.interpolate(coarse_t, size=(H, W), mode="bilinear", align_corners=False)
    noise = torch.from_numpy(rng.random((B * V, 3, H, W), dtype=np.float32))
    imgs = (0.7 * smooth + 0.3 * noise).reshape(B, V, 3, H, W).contiguous()
    out = make_cameras(B, V, H, W, depth_min=depth_min, depth_max=depth_max, behind_view=behind_view, rig=rig)
    out["imgs"] = imgs
    return out


def make_photo_case(B: int, V: int, H: int, W: int, *, seed: int = 0, behind_view: int = -1) -> Dict[str, torch.Tensor]:
    """Inputs of the unsupervised photometric loss at the loss resolution: the scene of ``make_scene`` plus one smooth depth
    map per view ``depths`` [V,B,H,W] (a slanted surface with bumps inside [depth_min, depth_max]; view v's map is what the
    network would predict with view v as the reference)."""
    out = make_scene(B, V, H, W, seed=seed, behind_view=behind_view)
    rng = np.random.default_rng(seed + 77)
    coarse = torch.from_numpy(rng.random((V * B, 1, 4, 5), dtype=np.float32))
    bumps = torch.nn.functional.interpolate(coarse, size=(H, W), mode="bicubic", align_corners=False).reshape(V, B, H, W)
    ramp = torch.linspace(0.0, 1.0, W).view(1, 1, 1, W)
    out["depths"] = (3.2 + 0.8 * ramp + 0.9 * (bumps - 0.5)).contiguous()
    return out


def make_features(B: int, V: int, C: int, h: int, w: int, *, seed: int = 1,
                  scale: float = 0.5) -> torch.Tensor:
    """Feature maps ~ N(0,1)*scale, [V,B,C,h,w] fp32 (hot-path-only timing input)."""
    rng = np.random.default_rng(seed)
    f = rng.standard_normal((V, B, C, h, w), dtype=np.float32) * np.float32(scale)
    return torch.from_numpy(f)


# --------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------
def _fan_in(shape: Sequence[int], transposed: bool) -> int:
    k = int(np.prod(shape[2:])) if len(shape) > 2 else 1
    cin = shape[0] if transposed else shape[1]
    return max(cin * k, 1)


def make_state_dict(template: Mapping[str, Tuple[int, ...]], *, seed: int = 0,
                    conv_gain: float = 1.6, head_gain: Mapping[str, float] | None = None,
                    transposed_keys: Sequence[str] = ()) -> "OrderedDict[str, torch.Tensor]":
    """Fill a ``{name: shape}`` template (a model's ``state_dict`` key/shape list)
    with deterministic sharpened weights.

    * conv / deconv weights ~ N(0, (gain / sqrt(fan_in))^2)
    * BatchNorm weight ~ U(0.6, 1.4), bias ~ N(0, 0.2), running_mean ~ N(0, 0.2),
      running_var ~ U(0.5, 1.5); ``num_batches_tracked`` = 1
    * conv biases ~ N(0, 0.1); scalar parameters (MVSNet-s ``temp``) = 1
    * ``head_gain`` maps a key substring to an extra multiplier (used to push the
      final 1-channel ``prob`` convs to logit std ~3 so that the softmax peaks).

    The draw order is the template's iteration order, so two models with the same
    key list get identical tensors.
    """
    rng = np.random.default_rng(seed)
    head_gain = dict(head_gain or {})
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in template.items():
        shape = tuple(int(s) for s in shape)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[name] = torch.tensor(1, dtype=torch.long)
            continue
        if len(shape) >= 3:  # convolution kernels (2D or 3D)
            transposed = any(name.startswith(k) or k in name for k in transposed_keys)
            std = conv_gain / math.sqrt(_fan_in(shape, transposed))
            for sub, g in head_gain.items():
                if sub in name:
                    std *= g
            arr = rng.standard_normal(shape, dtype=np.float32) * np.float32(std)
        elif leaf == "running_var":
            arr = rng.uniform(0.5, 1.5, size=shape).astype(np.float32)
        elif leaf == "running_mean":
            arr = (rng.standard_normal(shape) * 0.2).astype(np.float32)
        elif leaf == "weight":  # BatchNorm gamma (1-D)
            arr = rng.uniform(0.6, 1.4, size=shape).astype(np.float32)
        elif leaf == "bias":
            arr = (rng.standard_normal(shape) * (0.2 if len(shape) else 0.1)).astype(np.float32)
        elif leaf == "temp":
            arr = np.ones(shape, dtype=np.float32)
        else:
            arr = (rng.standard_normal(shape) * 0.1).astype(np.float32)
        out[name] = torch.from_numpy(np.ascontiguousarray(arr))
    return out


def template_of(module: torch.nn.Module) -> "OrderedDict[str, Tuple[int, ...]]":
    """``{name: shape}`` of a module's state dict, in registration order."""
    return OrderedDict((k, tuple(v.shape)) for k, v in module.state_dict().items())


# gains found by probing the reference on CPU (tests/golden/gen_golden.py prints the
# resulting softmax peak): they give max_d p in the 0.3-0.9 band on make_scene inputs.
SHARPEN = {
    "mvsnet": dict(conv_gain=1.414, head_gain={"feature.": 0.85, "cost_regularization.prob.weight": 10.0}),
    "vis": dict(conv_gain=1.0, head_gain={"final_conv.weight": 2.0}),
    "cvp": dict(conv_gain=1.2, head_gain={"prob0.weight": 12.0}),
}
TRANSPOSED_KEYS = {
    "mvsnet": ("cost_regularization.conv7.0", "cost_regularization.conv9.0", "cost_regularization.conv11.0"),
    "vis": ("dec_blocks",),
    "cvp": ("cost_reg_refine.conv5.0", "cost_reg_refine.conv6.0"),
}


def sharpened_state_dict(arch: str, template: Mapping[str, Tuple[int, ...]], seed: int = 0):
    """``arch`` in {"mvsnet", "vis", "cvp"}."""
    return make_state_dict(template, seed=seed, transposed_keys=TRANSPOSED_KEYS[arch], **SHARPEN[arch])


# ---- training fixtures (shared by tests/golden/gen_golden.py, the oracle tests and the GPU tests) ------------------
TRAIN_PROB_GAIN = 0.3   # keeps the softmax unsaturated under batch-statistics BatchNorm (mean max-prob ~0.5)


def train_state_dict(arch: str, template: Mapping[str, Tuple[int, ...]], seed: int = 0):
    """Sharpened weights for a train()-mode step: ``sharpened_state_dict`` with the `prob` head scaled down."""
    sd = sharpened_state_dict(arch, template, seed=seed)
    for k in list(sd.keys()):
        if k.endswith("cost_regularization.prob.weight") or k.endswith("cost_reg_refine.prob0.weight"):
            sd[k] = sd[k] * TRAIN_PROB_GAIN
    return sd


def train_target(scene, h: int, w: int, seed: int = 11):
    """Seeded ground-truth depth [B,h,w] and validity mask of the supervised loss (models/trainer.py:163-167)."""
    g = torch.Generator().manual_seed(seed)
    dmin, dmax = scene["depth_min"][:, 0].view(-1, 1, 1), scene["depth_max"][:, 0].view(-1, 1, 1)
    gt = dmin + (dmax - dmin) * (0.2 + 0.6 * torch.rand(dmin.shape[0], h, w, generator=g))
    mask = (torch.rand(dmin.shape[0], h, w, generator=g) > 0.1).float()
    return gt, mask


def supervised_loss(depth: torch.Tensor, gt: torch.Tensor, mask: torch.Tensor, depth_min: torch.Tensor, depth_max: torch.Tensor):
    """sum(|d - gt| / interval * mask) / sum(mask), interval = (max - min) / 128 of view 0 (models/trainer.py:163-167)."""
    interval = ((depth_max - depth_min) / 128)[:, 0].view(-1, 1, 1)
    return torch.sum(torch.abs(depth - gt) / interval * mask) / torch.sum(mask)


def supervised_loss_list(depth_list: Sequence[torch.Tensor], gt: torch.Tensor, mask: torch.Tensor, depth_min: torch.Tensor,
                         depth_max: torch.Tensor, reference_frame: int = 0):
    """The supervised loss of models/trainer.py:118-167 over ``depth_est_list``: ground truth and mask are bilinearly
    resized to each estimate (the mask keeps only pixels whose four neighbours are valid), every level has factor 1."""
    import torch.nn.functional as F
    interval = ((depth_max - depth_min) / 128)[:, reference_frame].view(-1, 1, 1)
    loss = 0
    for d in depth_list:
        if d is None:
            continue
        hd, wd = d.shape[1:]
        g = F.interpolate(gt.unsqueeze(1), size=(hd, wd), mode="bilinear", align_corners=False).squeeze(1)
        m = (F.interpolate(mask.unsqueeze(1).float(), size=(hd, wd), mode="bilinear", align_corners=False).squeeze(1) == 1).float()
        loss = loss + torch.sum(torch.abs(d - g) / interval * m) / torch.sum(m)
    return loss


VIS_LOSS_FACTORS = (2, 1, 0.5)   # models/trainer.py:33


def vis_supervised_loss(out, gt: torch.Tensor, mask: torch.Tensor, depth_min: torch.Tensor, depth_max: torch.Tensor, n_views: int):
    """The supervised Vis-MVSNet loss of models/trainer.py:118-206: per stage ``factor * masked L1`` on the fused depth plus
    ``factor / (n - 1) * bayesian_version_loss`` (models/utils.py:110-119) on every pair depth with its log-uncertainty."""
    import torch.nn.functional as F
    interval = ((depth_max - depth_min) / 128)[:, 0].view(-1, 1, 1, 1)
    loss = 0
    for i, d in enumerate(out["depth_est_list"]):
        hd, wd = d.shape[1:]
        g = F.interpolate(gt.unsqueeze(1), size=(hd, wd), mode="bilinear", align_corners=False)
        m = (F.interpolate(mask.unsqueeze(1).float(), size=(hd, wd), mode="bilinear", align_corners=False) == 1).float()
        l1 = torch.abs(d.unsqueeze(1) - g) / interval
        loss = loss + VIS_LOSS_FACTORS[i] * torch.sum(l1 * m) / torch.sum(m)
        for dp, (unc,) in out["depth_pair_list"][i]:
            l1p = torch.abs(dp.squeeze(1).unsqueeze(1) - g) / interval
            loss = loss + VIS_LOSS_FACTORS[i] / (n_views - 1) * (torch.sum((l1p * torch.exp(-unc) + unc) * m) / torch.sum(m)
                                                               + torch.sum(l1p * m) / torch.sum(m))
    return loss
