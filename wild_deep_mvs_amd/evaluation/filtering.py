"""Geometric-consistency filtering on MI355X -- drop-in for ``evaluation/filtering.py`` of fdarmon/wild_deep_mvs.

``run(dataloader, args)`` keeps the reference's interface (``evaluation/filtering.py:25-91``): it reads the depth maps
that ``run_depthmaps`` wrote under ``<data_path>/IntRes/depthmaps/<model>_<nviews>/<scene>/`` and writes
``mask_depth``, ``mask_disp`` and ``geo_mask`` to ``<data_path>/IntRes/geometric_filtering/...``.  The per-image work
(lines 60-83 of the reference: a CPU chain of point-cloud tensors) is one HIP launch, ``pscv_geo_filter``.
``geometric_masks`` is the same step as a function from tensors to masks.
"""
from __future__ import annotations

from pathlib import Path
from typing import Sequence

import numpy as np
import torch
import torch.nn.functional as F

from .. import ops


def depth_folder_name(args) -> str:
    """``evaluation/pipeline_utils.py:83-85``."""
    return f"{args.model}_{args.nviews}"


def geometric_masks(depth: torch.Tensor, src_depth: Sequence[torch.Tensor], K: torch.Tensor, R: torch.Tensor,
                    t: torch.Tensor, *, max_reproj_error: float = 1.0, depth_threshold: float = 0.01,
                    min_tri_angle: float = 1.0, num_consistent: int = 3, device="cuda"):
    """depth [h,w], src_depth N x [h_i,w_i], K,R [N+1,3,3], t [N+1,3,1] (view 0 = reference; intrinsics at the depth
    maps' resolution) -> (mask_depth, mask_disp, geo_mask), bool [h,w] on ``device``.  Defaults are the reference's
    command-line defaults (``pipeline_utils.py:49-52``)."""
    cams = ops.geo_filter_cams(K, R, t).to(device)
    return ops.geo_filter(depth.to(device), [s.to(device) for s in src_depth], cams, max_reproj_error=max_reproj_error,
                          depth_threshold=depth_threshold, min_tri_angle=min_tri_angle, num_consistent=num_consistent)


def run(dataloader, args):
    folder_name = depth_folder_name(args)
    out = Path(args.data_path) / "IntRes" / "geometric_filtering" / folder_name / str(args.scene)
    if (out / "finished.txt").exists():
        print("Filtering already done")
        return
    out.mkdir(parents=True, exist_ok=True)
    depth_folder = Path(args.data_path) / "IntRes" / "depthmaps" / folder_name / str(args.scene)

    for batch in dataloader:
        filename = batch["filename"][0]
        K, R, t = batch["K"][0].clone(), batch["R"][0], batch["t"][0]
        depth = torch.from_numpy(np.load(depth_folder / f"{filename}_out.npz")["depthmap"])
        src_depth = [torch.from_numpy(np.load(depth_folder / f"{f[0]}_out.npz")["depthmap"]) for f in batch["src_filenames"]]
        downscale = 1 if args.upsample else args.downscale          # filtering.py:51-52
        K[:, :2] /= downscale
        if args.upsample:                                            # filtering.py:54-58 (nearest, like the reference)
            h, w = depth.shape
            depth = F.interpolate(depth.view(1, 1, h, w), scale_factor=args.downscale).squeeze()
            src_depth = [F.interpolate(d.unsqueeze(0).unsqueeze(0), scale_factor=args.downscale).squeeze() for d in src_depth]
        with torch.no_grad():
            mask_depth, mask_disp, geo_mask = geometric_masks(
                depth, src_depth, K, R, t, max_reproj_error=args.max_reproj_error, depth_threshold=args.depth_threshold,
                min_tri_angle=args.min_tri_angle, num_consistent=args.num_consistent)
        np.savez_compressed(out / f"{filename}_out.npz", mask_depth=mask_depth.cpu().numpy(),
                            mask_disp=mask_disp.cpu().numpy(), geo_mask=geo_mask.cpu().numpy())
        if args.debug:
            return
    with open(out / "finished.txt", "a") as f:
        f.write(" ")
