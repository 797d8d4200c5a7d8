"""Drop-in for the reference's ``models/trainer.py``: the same class with the same methods (``step``, ``test``,
``forward_network``, ``loss``, ``photometricloss``, ``masked_photometricloss``, ``get_flow_from_depthmap`` and, inherited from the
harness base, ``keep_losses`` / ``log_iter`` / ``log_epoch``), so the reference's ``train.py`` drives it unchanged when this
package is installed as ``models``.  The loss half -- depth map -> flows -> warped source images ->
SSIM, with the gradient flowing back to the depth map -- runs on the pscv HIP kernels; ``step`` / ``test`` are host-side
orchestration around ``self.model(...)`` (resizing, loss weighting, metric bookkeeping) restated from trainer.py:61-207,280-321.

One ``pscv_photo_warp`` launch warps ALL source views (the reference loops over views with ``F.grid_sample``), one
``pscv_ssim`` launch compares them all with the reference image; backward is ``pscv_ssim_bwd`` + ``pscv_photo_warp_bwd``."""
from __future__ import annotations

import torch
import torch.distributed as dist

import torch.nn.functional as F

from .. import ops
from .. import training as T
from ..utils.ssimLoss import SSIM
from ..utils.trainer import Trainer as _HarnessTrainer
from ..utils.utils_3D import build_grid, build_proj_matrices, flows_from_single_depthmap, normalize  # noqa: F401
from .utils import *  # noqa: F401,F403  (the reference's module namespace: trainer.py:16-19)
from .utils import AbsDepthError_metrics, Thres_metrics, bayesian_version_loss, rec_upsample, tocuda


def _resize_views(imgs, size):
    """[b,n,c,h,w] -> [b,n,c,*size], bilinear, ``align_corners=False`` (trainer.py:65-68,108-110)."""
    b, n, c, h, w = imgs.shape
    return F.interpolate(imgs.reshape(b * n, c, h, w), size=size, mode="bilinear", align_corners=False).view(b, n, c, *size)


def _masked_mean_strict(values, mask):
    """sum(values mask) / sum(mask) without the empty-mask guard (the supervised branch, trainer.py:157)."""
    return torch.sum(values * mask) / torch.sum(mask)


def _masked_mean(values, mask):
    """sum(values mask) / sum(mask); the un-normalised sum (= 0, graph kept) when the mask is empty (trainer.py:159-163)."""
    total, count = torch.sum(values * mask), torch.sum(mask)
    return total / count if count != 0 else total


class Trainer(_HarnessTrainer):
    """``args`` fields used: ``architecture``, ``upsample_training``, ``occ_masking``, ``geom_clamping``, ``supervised``,
    ``num_im_train``, ``print_every``, ``dataset`` (trainer.py:26-51).  ``model`` / ``args`` may be omitted when only the loss
    methods are used (tests, function-level callers)."""

    def __init__(self, model=None, args=None):
        super().__init__()
        self.model, self.args = model, args
        self.ssim = SSIM()
        self.group = None        # process group of masked_photometricloss's all_gather (None = default group)
        self.factors_loss = [2, 1, 0.5]                 # Vis-MVSNet cascade weights, coarse to fine
        arch = getattr(args, "architecture", "") or ""
        upsample = bool(getattr(args, "upsample_training", False))
        # inputs are shrunk before the network when training with upsampling (CVP: 1/4, Vis: 1/2) ...
        self.input_down = {"cvp_mvsnet": 4, "vis_mvsnet": 2}.get(arch, 1) if upsample else 1
        # ... and the loss is taken at full resolution then; otherwise at the architecture's output resolution
        self.output_down = 1 if upsample else (4 if arch.startswith("mvsnet") else 2 if arch == "vis_mvsnet" else 1)

    # ---- network call + logged images (trainer.py:61-94) ------------------------------------------------------------------
    def forward_network(self, cuda_sample, ref_idx):
        imgs = cuda_sample["imgs"]
        b, n, c, h, w = imgs.shape
        src_idx = [i for i in range(self.args.num_im_train) if i != ref_idx]
        K = cuda_sample["K"].clone()
        K[:, :, :2] /= self.input_down
        outputs = self.model(_resize_views(imgs, (h // self.input_down, w // self.input_down)), K, cuda_sample["R"],
                             cuda_sample["t"], cuda_sample["depth_min"], cuda_sample["depth_max"], reference_frame=ref_idx)
        self.ims = {"ref_img": imgs[:, 0]}
        for k, i in enumerate(src_idx):
            self.ims[f"src_img_{k}"] = imgs[:, i]
        lo = cuda_sample["depth_min"][:, 0].view(-1, 1, 1)
        hi = cuda_sample["depth_max"][:, 0].view(-1, 1, 1)
        for k, d in enumerate(outputs["depth_est_list"]):
            if d is not None:
                self.ims[f"scale_{k}_depth_est"] = torch.clamp((d.detach() - lo) / (hi - lo), 0, 1).unsqueeze(1).expand(-1, 3, -1, -1)
        return outputs

    # ---- one training / validation iteration (trainer.py:96-206) ----------------------------------------------------------------
    def step(self, sample, train):
        cuda_sample = tocuda(sample)
        b, n, c, h, w = cuda_sample["imgs"].shape
        vis = self.args.architecture == "vis_mvsnet"
        ref_idx = dist.get_rank() if self.args.occ_masking else 0      # occlusion masking: rank r predicts view r
        src_idx = [i for i in range(self.args.num_im_train) if i != ref_idx]
        outputs = self.forward_network(cuda_sample, ref_idx)
        out_hw = (h // self.output_down, w // self.output_down)
        img = _resize_views(cuda_sample["imgs"], out_hw)

        if self.args.supervised:
            depth_list, pairs_list = outputs["depth_est_list"], outputs["depth_pair_list"]
            gt, gt_mask = sample["depth"].cuda(), sample["mask"].cuda().float()
            gts, masks = [], []
            for d in depth_list:
                if d is None:
                    gts.append(None); masks.append(None)
                    continue
                gts.append(F.interpolate(gt, size=tuple(d.shape[1:]), mode="bilinear", align_corners=False))
                # exactly 1 after bilinear resizing: all four neighbours carry a valid depth
                masks.append((F.interpolate(gt_mask, size=tuple(d.shape[1:]), mode="bilinear", align_corners=False) == 1).float())
            interval = ((cuda_sample["depth_max"] - cuda_sample["depth_min"]) / 128)[:, 0].view(b, 1, 1, 1)
        else:
            depth_list = rec_upsample(outputs["depth_est_list"], out_hw)
            pairs_list = rec_upsample(outputs["depth_pair_list"], out_hw)
            K = cuda_sample["K"].clone()
            K[:, :, :2] /= self.output_down
            proj_mat = build_proj_matrices(K, cuda_sample["R"], cuda_sample["t"])

        loss = 0
        for k, d in enumerate(depth_list):
            if d is None:
                continue
            factor = self.factors_loss[k] if vis else 1
            if self.args.supervised:
                loss = loss + factor * _masked_mean_strict(torch.abs(d.unsqueeze(1) - gts[k]) / interval, masks[k])
            else:
                ssim, mask = self.loss(img, d, proj_mat, idxs=None, suffix=f"_scale{k}")
                loss = loss + factor * _masked_mean(ssim, mask)

        for k, pairs in enumerate(pairs_list):
            factor = (self.factors_loss[k] if vis else 1) / (n - 1)
            for j, (d, (unc,)) in enumerate(pairs):
                if d is None:
                    continue
                d = d.squeeze(1)
                if self.args.supervised:
                    loss = loss + factor * bayesian_version_loss(torch.abs(d.unsqueeze(1) - gts[k]) / interval, unc, masks[k])
                else:
                    # no occlusion masking here: the confidence head has to see occlusions to learn them
                    pair = [ref_idx, src_idx[j]]
                    ssim, mask = self.photometricloss(img[:, pair], d, proj_mat[:, pair], suffix=f"_scale{k}_pairwise{j}")
                    loss = loss + factor * bayesian_version_loss(ssim, unc, mask)

        self.keep_losses({("train_loss" if train else "val_loss"): loss.detach()})
        self.nb_iter += 1
        return loss

    # ---- evaluation iteration (trainer.py:280-321) ---------------------------------------------------------------------------
    def test(self, sample):
        cuda_sample = tocuda(sample)
        first = (lambda x: x[:, 0]) if isinstance(sample["imgs"], torch.Tensor) else (lambda x: x[0])
        mask, depth_gt = first(cuda_sample["mask"]), first(cuda_sample["depth"])
        extra = {}
        if self.args.architecture == "vis_mvsnet":
            extra = dict(depth_nums=[64, 32, 16], scales=[2, 1, 0.5])     # twice the training plane count (``scales`` is ignored
                                                                          # by the model, as in the reference: frontend.py:33-41)
        elif self.args.architecture == "cvp_mvsnet" and self.args.dataset != "dtu_yao":
            extra = dict(nscale=4)
        with torch.no_grad():
            outputs = self.model(cuda_sample["imgs"], cuda_sample["K"], cuda_sample["R"], cuda_sample["t"],
                                 cuda_sample["depth_min"], cuda_sample["depth_max"], **extra)
            h, w = mask.shape[-2:]
            step = ((cuda_sample["depth_max"] - cuda_sample["depth_min"]) / 128)[:, 0]
            est = F.interpolate(outputs["depth"].unsqueeze(1), (h, w), mode="bilinear", align_corners=False).squeeze(1) / step
            gt = depth_gt / step
        valid = mask > 0.5
        self.keep_losses({"EPE": AbsDepthError_metrics(est, gt, valid).detach(),
                          "1pxError": Thres_metrics(est, gt, valid, 1).detach(),
                          "3pxError": Thres_metrics(est, gt, valid, 3).detach()})
        self.nb_iter += 1

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
