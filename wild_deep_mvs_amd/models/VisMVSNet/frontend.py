"""Drop-in for the reference's ``models/VisMVSNet/frontend.py``: common ``forward()`` -> 3-stage Vis cascade."""
import os

import torch
import torch.distributed
from torch import nn
from torch.nn import functional as F

from ...graph import ReplayHooks, replayable
from .model_cas import Model


class Frontend(ReplayHooks, nn.Module):
    def __init__(self):
        super().__init__()
        self.model = Model()
        self.depth_nums = [32, 16, 8]
        self.interval_scales = [4, 2, 1]
        # 2-D extractor: "pscv" = the residual U-Net on MFMA conv2d launches writing the warp kernel's channels-last 16-bit
        # layout directly (default); "torch" = PyTorch-ROCm in fp32, converted where the warp kernel reads it
        self.feature_engine = "pscv"
        # the 2-D extractor in train(): "torch" = PyTorch-ROCm autograd in fp32 (default); "pscv" = FeatExt.forward_train: all views in one
        # engine pass (grouped BatchNorm statistics), 16-bit activations
        self.feature_engine_train = os.environ.get("PSCV_FEATURE_ENGINE_TRAIN", "torch")

    @property
    def storage_dtype(self):
        return self.model.stage1.storage_dtype

    @storage_dtype.setter
    def storage_dtype(self, dt):
        for st in (self.model.stage1, self.model.stage2, self.model.stage3):
            st.storage_dtype = dt

    @property
    def train_storage_dtype(self):
        return self.model.stage1.train_storage_dtype

    @train_storage_dtype.setter
    def train_storage_dtype(self, dt):
        for st in (self.model.stage1, self.model.stage2, self.model.stage3):
            st.train_storage_dtype = dt

    def set_view_group(self, group):
        """Shard the source views of every stage over a torch.distributed group (None = no sharding)."""
        for st in (self.model.stage1, self.model.stage2, self.model.stage3):
            st.view_group = group

    def set_depth_group(self, group):
        """Shard the depth planes of every stage over a torch.distributed group (None = no sharding)."""
        for st in (self.model.stage1, self.model.stage2, self.model.stage3):
            st.depth_group = group

    feature_shard_group = None     # see set_depth_row_groups

    def set_row_group(self, group, stages=(2, 3)):
        """Shard the image ROWS of the given cascade stages (1-based; default: the per-pixel stages 2 and 3, whose 32 / 16 planes
        leave nothing to shard along depth) over a torch.distributed group; the other stages are left as they are.  Combined with
        ``set_stage_depth_group(group, stages=(1,))`` no stage of configuration 3 runs replicated."""
        for k, st in enumerate((self.model.stage1, self.model.stage2, self.model.stage3), start=1):
            if k in stages:
                st.row_group = group

    def set_depth_row_groups(self, group):
        """The shard of configuration 3 with no replicated stage (round 4): stage 1 (64-192 fronto-parallel planes) by depth planes,
        the per-pixel stages 2-3 (32 / 16 planes) by image rows.  ``None`` removes both."""
        self.set_stage_depth_group(group, stages=(1,))
        self.set_row_group(group, stages=(2, 3))
        self.feature_shard_group = group          # the 2-D extractor too: rank r extracts views r, r + G, ...; one all-gather per scale

    def set_stage_depth_group(self, group, stages=(1,)):
        """``set_depth_group`` for selected stages only (1-based)."""
        for k, st in enumerate((self.model.stage1, self.model.stage2, self.model.stage3), start=1):
            if k in stages:
                st.depth_group = group

    def fill_cam_array(self, K, R, t, start_depth, depth_interval):
        b = K.shape[0]
        cam = torch.zeros((b, 2, 4, 4), device=K.device)
        cam[:, 0, :3, :3], cam[:, 0, :3, 3:4], cam[:, 1, :3, :3] = R, t, K
        cam[:, 1, 3, 0], cam[:, 1, 3, 1] = start_depth, depth_interval
        return cam

    def fill_cam_arrays(self, K, R, t, start_depth, depth_interval):
        """``fill_cam_array`` of every view in six launches: [V,b,2,4,4], view-major so that ``[1:].unbind(0)`` hands the stages
        their source cameras as back-to-back views of one buffer (``ops.homog_cams_device`` then reads them without a stack)."""
        b, v = K.shape[:2]
        cam = torch.zeros((v, b, 2, 4, 4), device=K.device)
        cam[:, :, 0, :3, :3], cam[:, :, 0, :3, 3:4], cam[:, :, 1, :3, :3] = R.transpose(0, 1), t.transpose(0, 1), K.transpose(0, 1)
        cam[:, :, 1, 3, 0], cam[:, :, 1, 3, 1] = start_depth.transpose(0, 1), depth_interval.transpose(0, 1)
        return cam

    @replayable
    def forward(self, imgs, K, R, t, depth_min, depth_max, reference_frame=0, **kwargs):
        depth_interval = (depth_max - depth_min) / 128                                   # frontend.py:27
        interval_scales = kwargs.get("interval_scales", self.interval_scales)
        depth_nums = kwargs.get("depth_nums", self.depth_nums)
        taps = kwargs.get("taps")
        imgs_all = None
        if not isinstance(imgs, (list, tuple)):
            imgs_all = imgs                                                              # [n,V,3,H,W]
            imgs = torch.unbind(imgs, dim=1)
        v = len(imgs)
        src_idx = [i for i in range(v) if i != reference_frame]
        n = imgs[reference_frame].shape[0]
        order = [reference_frame] + src_idx
        pick = (lambda a: torch.stack([a[:, i] for i in order], 1)) if reference_frame != 0 else (lambda a: a)   # (device-side only)
        cams = self.fill_cam_arrays(pick(K), pick(R), pick(t), pick(depth_min), pick(depth_interval))    # reference view first
        ref_cam, srcs_cam = cams[0], list(cams[1:].unbind(0))
        with torch.set_grad_enabled(self.training):
            grp = self.model.stage1.view_group
            if self.training:
                # train(): per-view extractor passes like the reference (frontend.py:59-61: each view normalises with its own
                # batch statistics), on PyTorch-ROCm autograd (upstream of the path); the stages are the engine's autograd nodes
                if grp is not None:
                    raise NotImplementedError("pscv Vis-MVSNet: the source-view shard is an inference path")
                if getattr(self, "feature_engine_train", "torch") == "pscv":
                    # the extractor of ALL views in one engine pass (FeatExt.forward_train: grouped BatchNorm statistics, one per view)
                    fdt = getattr(self, "feature_engine_train_dtype", None) or torch.float16     # the extractor's own 16-bit format (fp16: see
                    maps = self.model.feat_ext.forward_train(torch.cat([imgs[i] for i in order], 0), v, fdt)   # MVSNet.forward), then the sweep's
                    if fdt != self.train_storage_dtype:
                        maps = [m.to(self.train_storage_dtype) for m in maps]
                    packs = [torch.chunk(f, v, 0) for f in maps]
                    ref_feats = tuple(p[0] for p in packs)
                    src_feats = [tuple(p[j + 1] for p in packs) for j in range(len(src_idx))]
                else:
                    ref_feats = self.model.feat_ext(imgs[reference_frame])
                    src_feats = [self.model.feat_ext(imgs[i]) for i in src_idx]
            engine = self.feature_engine == "pscv" and not self.training
            fe = (lambda x: self.model.feat_ext.forward_engine(x, self.storage_dtype)) if engine else self.model.feat_ext
            for st in (self.model.stage1, self.model.stage2, self.model.stage3):
                st._channels_last_features = engine
            fgrp = getattr(self, "feature_shard_group", None)
            if self.training:
                pass
            elif (fgrp is not None and grp is None and engine and len({tuple(i.shape) for i in imgs}) == 1
                  and torch.distributed.get_world_size(fgrp) <= v):
                # (more ranks than views -- e.g. 8 ranks, 5 views: EVERY rank takes the replicated extractor below instead; the
                #  decision depends on (world, V) only, so all ranks agree before any collective is entered)
                # depth / row shards need every view's maps on every rank: extract V / G views here, all-gather the rest (16-bit
                # channels-last maps, 6.9 MB per view at 512x640) instead of running the whole extractor G times over
                import torch.distributed as dist
                world, rank = dist.get_world_size(fgrp), dist.get_rank(fgrp)
                slots = (v + world - 1) // world
                mine = [j for j in range(v) if j % world == rank]
                maps = fe(torch.cat([imgs[order[j]] for j in mine], 0))      # world <= V: every rank owns at least one view
                per_scale = []
                for k in range(3):
                    shp = (slots * n,) + tuple(maps[k].shape[1:])
                    buf = maps[k].new_zeros(shp)
                    buf[:len(mine) * n] = maps[k]
                    got = [torch.empty_like(buf) for _ in range(world)]
                    dist.all_gather(got, buf.contiguous(), group=fgrp)
                    per_scale.append([got[j % world][(j // world) * n:(j // world + 1) * n] for j in range(v)])
                ref_feats = tuple(per_scale[k][0] for k in range(3))
                src_feats = [tuple(per_scale[k][j + 1] for k in range(3)) for j in range(len(src_idx))]
            elif grp is None and len({tuple(i.shape) for i in imgs}) == 1:
                # all views through the 2-D extractor as one batch (same result as the per-view loop in eval mode)
                if imgs_all is not None and reference_frame == 0:
                    batch = imgs_all.transpose(0, 1).reshape((v * n,) + tuple(imgs_all.shape[2:]))   # view-major; a view when n == 1
                else:
                    batch = torch.cat([imgs[i] for i in order], 0)
                packs = [torch.chunk(f, v, 0) for f in fe(batch)]
                ref_feats = tuple(p[0] for p in packs)
                src_feats = [tuple(p[j + 1] for p in packs) for j in range(len(src_idx))]
            elif grp is None:
                ref_feats = fe(imgs[reference_frame])
                src_feats = [fe(imgs[i]) for i in src_idx]
            else:   # view shard: a rank only extracts the features of the source views it will sweep
                import torch.distributed as dist
                ref_feats = fe(imgs[reference_frame])
                world, rank = dist.get_world_size(grp), dist.get_rank(grp)
                src_feats = [fe(imgs[i]) if j % world == rank else (None, None, None)
                             for j, i in enumerate(src_idx)]
            di = depth_interval[:, reference_frame].view(n, 1, 1, 1)
            stages = (self.model.stage1, self.model.stage2, self.model.stage3)
            ests, probs, pairs = [], [], []
            start = None
            for k, (stage, s_scale) in enumerate(zip(stages, (8, 4, 2))):
                if k > 0:
                    # NB: like the reference, the offset uses the ATTRIBUTE self.interval_scales, not the kwarg
                    # (frontend.py:76-78,89-91)
                    cl_feats = engine or (self.training and self.feature_engine_train == "pscv")      # channels-last maps [n,h,w,C]
                    fhw = tuple(ref_feats[k].shape[1:3]) if cl_feats else tuple(ref_feats[k].shape[2:])
                    start = F.interpolate(ests[-1].detach(), size=fhw, mode='bilinear',
                                          align_corners=False) - depth_nums[k] * di * self.interval_scales[k] / 2
                stage_taps = {} if taps is not None else None
                est, prob, pr = stage([ref_feats[k], ref_cam, [f[k] for f in src_feats], srcs_cam], depth_num=depth_nums[k],
                                      upsample=False, mem=False, mode="soft", depth_start_override=start,
                                      depth_interval_override=di * interval_scales[k], s_scale=s_scale, taps=stage_taps)
                ests.append(est), probs.append(prob), pairs.append(pr)
                if taps is not None:
                    taps.setdefault("stages", []).append(stage_taps)
        prob_1_up = F.interpolate(probs[0], scale_factor=4, mode='bilinear', align_corners=False)
        prob_2_up = F.interpolate(probs[1], scale_factor=2, mode='bilinear', align_corners=False)
        return {
            "depth": ests[2].squeeze(1),
            "depth_est_list": [ests[2].squeeze(1), ests[1].squeeze(1), ests[0].squeeze(1)],
            "depth_pair_list": [pairs[2], pairs[1], pairs[0]],
            "photometric_confidence": torch.cat([prob_1_up, prob_2_up, probs[2]], dim=1),
        }
