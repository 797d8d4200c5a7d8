"""Drop-in for the loss half of the reference's ``models/trainer.py``: the unsupervised photometric
loss -- depth map -> flows -> warped source images -> SSIM, with the gradient flowing back to the depth map -- on the pscv HIP
kernels.  The rest of the reference's ``Trainer`` (optimiser stepping, logging, validation metrics) is the training harness
and stays the caller's (out of scope).

One ``pscv_photo_warp`` launch warps ALL source views (the reference loops over views with ``F.grid_sample``), one
``pscv_ssim`` launch compares them all with the reference image; backward is ``pscv_ssim_bwd`` + ``pscv_photo_warp_bwd``."""
from __future__ import annotations

import torch
import torch.distributed as dist

from .. import ops
from .. import training as T
from ..utils.ssimLoss import SSIM


class Trainer:
    """Holds what the loss methods of the reference's ``Trainer`` use: ``args`` (``occ_masking``, ``geom_clamping``), ``ssim``
    and the ``ims`` dict the reference logs warped images into (trainer.py:26-31, 53-58)."""

    def __init__(self, model=None, args=None):
        self.model, self.args = model, args
        self.ssim = SSIM()
        self.ims = {}
        self.group = None        # process group of masked_photometricloss's all_gather (None = default group)

    def loss(self, imgs, d, proj_mat, idxs, suffix=""):                                   # trainer.py:53-58
        if getattr(self.args, "occ_masking", False):
            return self.masked_photometricloss(imgs, d, proj_mat, idxs, suffix)
        return self.photometricloss(imgs, d, proj_mat, suffix)

    @staticmethod
    def _split(proj_mat, ref_idx):
        N = proj_mat.shape[1]
        src_idx = list(range(ref_idx)) + list(range(ref_idx + 1, N))
        pm = proj_mat.detach().to(torch.float32)
        sel = torch.stack([pm[:, i] for i in src_idx], dim=1).contiguous()
        return src_idx, ops.inv_proj4x4(pm[:, ref_idx]).contiguous(), sel

    def get_flow_from_depthmap(self, depth_est, proj_mat, src_size, ref_idx):
        """-> flows [b,N-1,h,w,2] ((size-1)-normalised, -10 behind the camera, clamped to +-10), depth in the source views
        [b,N-1,h,w] (trainer.py:209-219).  Function-level helper, no autograd: inside the loss the flows are never
        materialised and the gradient is produced by ``pscv_photo_warp_bwd``."""
        h, w = src_size
        if tuple(depth_est.shape[-2:]) != (int(h), int(w)):
            raise ValueError("pscv get_flow_from_depthmap: the source size must equal the depth map's (as in the reference's calls)")
        _, inv_ref, proj_src = self._split(proj_mat, ref_idx)
        o = ops.photo_warp(None, depth_est.detach().to(torch.float32), inv_ref, proj_src, want_mask=False, want_z=True, want_flows=True)
        return o["flows"], o["z"]

    def photometricloss(self, imgs, depth_est, proj_mat, suffix=""):
        """imgs [b,N,3,h,w], depth_est [b,h,w], proj_mat [b,N,4,4] -> (ssim [b,N-1,h,w], mask [b,N-1,h,w] fp32); view 0 is the
        reference (trainer.py:221-238)."""
        b, N, c, h, w = imgs.shape
        src_idx, inv_ref, proj_src = self._split(proj_mat, 0)
        imgs = imgs.detach().to(torch.float32)
        src = torch.stack([imgs[:, i] for i in src_idx], dim=1).contiguous()
        warped, mask = T.PhotoWarpFn.apply(depth_est.to(torch.float32), src, inv_ref, proj_src)
        ssim = self.ssim(imgs[:, 0].contiguous(), warped.reshape(b * (N - 1), c, h, w)).view(b, N - 1, c, h, w).mean(dim=2)
        for k, i in enumerate(src_idx):
            self.ims[f"warped{i}{suffix}"] = torch.clamp(warped[:, k].detach(), 0., 1.)
        return ssim, mask

    def masked_photometricloss(self, imgs, depth_est, proj_mat, idxs=None, suffix=""):
        """Occlusion-masked variant (trainer.py:240-278): rank r predicts the depth of view r; the maps are all-gathered and a
        source pixel only counts where its own depth map agrees with the reprojected reference depth.  -> (ssim [b,N-1,h,w],
        mask [b,N-1,h,w] bool)."""
        b, N, c, h, w = imgs.shape
        all_depthmaps = [torch.ones_like(depth_est) for _ in range(N)]
        dist.all_gather(all_depthmaps, depth_est.detach().contiguous(), group=self.group)
        i_ref = dist.get_rank(self.group)
        src_idx, inv_ref, proj_src = self._split(proj_mat, i_ref)
        imgs = imgs.detach().to(torch.float32)
        src = torch.stack([imgs[:, i] for i in src_idx], dim=1).contiguous()
        ref_depth = depth_est.squeeze(1).to(torch.float32)
        src_depth = torch.stack([all_depthmaps[i].squeeze(1).to(torch.float32) for i in src_idx], dim=1).contiguous()
        self.ims[f"warped{suffix}_ref_{i_ref}src_{i_ref}"] = torch.clamp(imgs[:, i_ref], 0., 1.)
        warped, mask = T.PhotoWarpFn.apply(ref_depth, src, inv_ref, proj_src)
        geo = ops.photo_warp(None, ref_depth.detach(), inv_ref, proj_src, src_depth=src_depth, want_mask=False, want_z=True)
        wsd = geo["warped_depth"]
        reproj_diff = torch.abs(geo["z"] - wsd) / torch.clamp(wsd, 1e-8)
        masks = (mask * (reproj_diff < self.args.geom_clamping)).bool()
        ssims = self.ssim(imgs[:, i_ref].contiguous(), warped.reshape(b * (N - 1), c, h, w)).view(b, N - 1, c, h, w).mean(dim=2)
        for k, i in enumerate(src_idx):
            self.ims[f"warped{suffix}_ref_{i_ref}src_{i}_masked"] = torch.clamp((masks[:, k].unsqueeze(1) * warped[:, k]).detach(), 0., 1.)
        return ssims, masks
