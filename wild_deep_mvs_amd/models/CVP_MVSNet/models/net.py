"""Drop-in for the reference's ``models/CVP_MVSNet/models/net.py`` on the pscv engine: feature pyramid (upstream,
PyTorch-ROCm), the CVP 3-D regulariser on MFMA conv launches, and the coarse-to-fine ``network.forward``."""
from __future__ import annotations

from typing import Dict, Optional

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import _lib as L
from .... import ops
from .... import training as T
from ...MVSNet.module import deconv_engine_layer
from .modules import (ConvBnReLU3D, calDepthHypo, calSweepingDepthHypo, conditionIntrinsics, conv, proj_cost, _cams)


class FeaturePyramid(nn.Module):
    """16-channel features at every pyramid level, finest first (reference net.py:21-47)."""
    _names = ("conv0aa", "conv0ba", "conv0bb", "conv0bc", "conv0bd", "conv0be", "conv0bf", "conv0bg", "conv0bh")
    _chans = ((3, 64), (64, 64), (64, 64), (64, 32), (32, 32), (32, 32), (32, 16), (16, 16), (16, 16))

    def __init__(self):
        super().__init__()
        for name, (ci, co) in zip(self._names, self._chans):
            setattr(self, name, conv(ci, co, kernel_size=3, stride=1))

    def _tower(self, x):
        for name in self._names:
            x = getattr(self, name)(x)
        return x

    def forward(self, img, scales=5):
        levels = [self._tower(img)]
        for _ in range(scales - 1):
            img = F.interpolate(img, scale_factor=0.5, mode='bilinear', align_corners=None).detach()
            levels.append(self._tower(img))
        return levels

    # -- HIP path: nine MFMA conv2d launches per level (bias + LeakyReLU(0.1) fused), channels-last 16-bit maps --
    def engine_layers(self, dtype: torch.dtype):
        key = (ops.weights_epoch(), dtype) + tuple((p.data_ptr(), p._version) for p in self.parameters())
        if getattr(self, "_layers", None) is None or self._layers_key != key:
            self._layers = [ops.Conv2dLayer.build(getattr(self, n)[0].weight, stride=1, conv_bias=getattr(self, n)[0].bias,
                                                  leaky=0.1, dtype=dtype) for n in self._names]
            self._layers_key = key
        return self._layers

    def forward_engine(self, img, scales: int, dtype: torch.dtype):
        """[B,3,H,W] on the GPU -> list of channels-last feature maps [B,H_l,W_l,16] in ``dtype``, finest first
        (image pyramid and input layout: ``ops.image_pyramid_cl8``)."""
        layers = self.engine_layers(dtype)

        def tower(y):
            for layer in layers:
                y = ops.conv2d(y, layer)
            return y
        # image pyramid (F.interpolate(img, scale_factor=0.5, mode='bilinear') between the levels, net.py:44) + the towers' 8-channel
        # 16-bit input layout: one launch per level (ops.image_pyramid_cl8, same bits as F.interpolate + the conversion)
        return [tower(y) for y in ops.image_pyramid_cl8(img, scales, dtype)]


class CostRegNet(nn.Module):
    """CVP regulariser (reference net.py:50-85) on the engine: [B,D,h,w,16] 16-bit -> fp32 logits [B,D,h,w]."""

    def __init__(self):
        super().__init__()
        self.conv0 = ConvBnReLU3D(16, 16, kernel_size=3, pad=1)
        self.conv0a = ConvBnReLU3D(16, 16, kernel_size=3, pad=1)
        self.conv1 = ConvBnReLU3D(16, 32, stride=2, kernel_size=3, pad=1)
        self.conv2 = ConvBnReLU3D(32, 32, kernel_size=3, pad=1)
        self.conv2a = ConvBnReLU3D(32, 32, kernel_size=3, pad=1)
        self.conv3 = ConvBnReLU3D(32, 64, kernel_size=3, pad=1)
        self.conv4 = ConvBnReLU3D(64, 64, kernel_size=3, pad=1)
        self.conv4a = ConvBnReLU3D(64, 64, kernel_size=3, pad=1)
        self.conv5 = nn.Sequential(nn.ConvTranspose3d(64, 32, kernel_size=3, padding=1, output_padding=0, stride=1, bias=False),
                                   nn.BatchNorm3d(32), nn.ReLU(inplace=True))
        self.conv6 = nn.Sequential(nn.ConvTranspose3d(32, 16, kernel_size=3, padding=1, output_padding=1, stride=2, bias=False),
                                   nn.BatchNorm3d(16), nn.ReLU(inplace=True))
        self.prob0 = nn.Conv3d(16, 1, 3, stride=1, padding=1)
        self._lay, self._key = None, None

    def engine_layers(self, dtype) -> Dict[str, ops.Conv3dLayer]:
        key = (ops.weights_epoch(), dtype) + tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))
        if self._lay is None or key != self._key:
            dev = self.prob0.weight.device
            lay = {n: getattr(self, n).engine_layer(dev, dtype)
                   for n in ("conv0", "conv0a", "conv1", "conv2", "conv2a", "conv3", "conv4", "conv4a")}
            lay["conv5"] = deconv_engine_layer(self.conv5, dev, stride=1, dtype=dtype)   # stride-1 deconv == flipped conv
            lay["conv6"] = deconv_engine_layer(self.conv6, dev, stride=2, dtype=dtype)
            lay["prob0"] = ops.Conv3dLayer.build(self.prob0.weight, kind=L.CONV_S1, device=dev, conv_bias=self.prob0.bias, dtype=dtype)
            self._lay, self._key = lay, key
        return self._lay

    def train_blocks(self):
        """The regulariser as the block list of the training executor (``training.RegressFn``): the dataflow of ``forward``
        (reference net.py:76-85) with batch-statistics BatchNorm."""
        blocks, prev = [], "cost"
        for name, stride in (("conv0", 1), ("conv0a", 1), ("conv1", 2), ("conv2", 1), ("conv2a", 1), ("conv3", 1), ("conv4", 1),
                             ("conv4a", 1)):
            m = getattr(self, name)
            blocks.append(T.Block(name, prev, m.conv.weight, stride=stride, bn=m.bn, relu=True))
            prev = name
        blocks.append(T.Block("conv5", prev, self.conv5[0].weight, stride=1, transposed=True, bn=self.conv5[1], relu=True, skip="conv2a"))
        blocks.append(T.Block("conv6", "conv5", self.conv6[0].weight, stride=2, transposed=True, bn=self.conv6[1], relu=True, skip="conv0a"))
        blocks.append(T.Block("prob0", "conv6", self.prob0.weight, bn=None, relu=False, conv_bias=self.prob0.bias))
        return blocks

    def forward(self, x: torch.Tensor, taps: Optional[dict] = None) -> torch.Tensor:
        if self.training:
            raise RuntimeError("pscv CVP CostRegNet: in train() mode the regulariser runs inside training.RegressFn "
                               "(network.forward routes there); this entry point is the eval-mode engine")
        B, D, h, w, _ = x.shape
        if D % 2 or h % 2 or w % 2:
            raise ValueError(f"CVP CostRegNet needs even D,h,w (got {D},{h},{w}), as in the reference")
        ly = self.engine_layers(x.dtype)
        c0 = ops.conv3d(ops.conv3d(x, ly["conv0"]), ly["conv0a"])
        c2 = ops.conv3d(ops.conv3d(ops.conv3d(c0, ly["conv1"]), ly["conv2"]), ly["conv2a"])
        c4 = ops.conv3d(ops.conv3d(ops.conv3d(c2, ly["conv3"]), ly["conv4"]), ly["conv4a"])
        c5 = ops.conv3d(c4, ly["conv5"], skip=c2)          # conv2 + relu(bn(deconv))   net.py:81
        c6 = ops.conv3d(c5, ly["conv6"], skip=c0)          # net.py:82
        logits = ops.conv3d(c6, ly["prob0"], out_dtype=torch.float32)
        if taps is not None:
            taps.update(conv0=c0, conv2=c2, conv4=c4, conv5=c5, conv6=c6)
        return logits.view(B, D, h, w)


class network(nn.Module):
    def __init__(self):
        super().__init__()
        self.featurePyramid = FeaturePyramid()
        self.cost_reg_refine = CostRegNet()
        self.nscale = 2
        self.storage_dtype = torch.float16
        # fronto-parallel planes of the coarsest level in eval mode: 96 like the reference (net.py:126-127, `48 if self.training else
        # 96`); BASELINE.json's configuration 4 names the train-mode count 48 -- set this to 48 to run that case in eval mode
        self.coarse_planes_eval = 96
        # 2-D pyramid tower: "pscv" = MFMA conv2d launches writing channels-last 16-bit maps (default);
        # "torch" = PyTorch-ROCm in fp32, converted where the warp kernel reads them
        self.feature_engine = "pscv"
        self.train_storage_dtype = torch.bfloat16   # train(): bf16 activations / gradients by default (range), fp32 accumulation
        # 2-D pyramid tower in train(): "torch" = PyTorch-ROCm autograd in fp32 (default, like MVSNet's extractor);
        # "pscv" = training.FeaturePyramidFn: all views in one engine pass, 16-bit activations
        self.feature_engine_train_dtype = None       # 16-bit format of the engine tower's activations in train() (None: fp16)
        self.feature_engine_train = os.environ.get("PSCV_FEATURE_ENGINE_TRAIN", "torch")
        # row-slab shard of the refinement levels (``Frontend.set_row_group``): with a torch.distributed group
        # set here, rank r builds, regularises and regresses image rows [ra, rb) of every refinement level (8 per-pixel planes x
        # H x W: 10.5 M voxels at 1024 x 1280, net.py:166-210) plus a recomputed ROW_HALO-row halo, and the ranks all-gather
        # their rows of the level's depth (+ confidence on the last level).  The pyramid tower, the camera blocks, the per-pixel
        # hypotheses (a pass over H x W pixels) and the small coarsest level stay replicated.  The reference has no counterpart.
        self.row_group = None

    # rows a refinement level's depth depends on beyond its own: the regulariser's receptive field in the image plane is +-17 voxels
    # (conv0, conv0a 2; conv1 1; five 3x3x3 layers at half resolution 10; conv5^T 2; conv6^T 1; prob0 1 -- net.py:50-85, probed in
    # SURVEY.md section 8e), rounded up to a multiple of 4: the stride-2 level then sees the slab in the phase it has in the image
    ROW_HALO = 20

    def _refine_level_row_shard(self, ref_map, src_maps, cams, hyp, want_conf):
        """One refinement level (reference net.py:166-219) with the image rows sharded over ``self.row_group``.  ``ref_map`` [B,H,W,16]
        channels-last, ``src_maps`` whole, ``cams`` of the whole image, ``hyp`` [B,8,H,W] per-pixel hypotheses.  The slab's cost
        volume is bit-identical to those rows of the whole-image launch (`pscv_warp_cost_rows`: slab row y is evaluated at
        (x, y + slab origin)); values ROW_HALO rows inside an artificial border equal the unsharded ones.  Returns the level's depth
        [B,H,W] (and confidence [B,H,W] or None) on every rank: one all-gather of the owned rows."""
        import torch.distributed as dist
        from .... import dist as pdist
        grp = self.row_group
        world, rank = dist.get_world_size(grp), dist.get_rank(grp)
        B, H, W, _ = ref_map.shape
        bounds = [pdist.plane_shard(H, world, r, multiple=4) for r in range(world)]
        ra, rb = bounds[rank]
        ea, eb = max(0, ra - self.ROW_HALO), min(H, rb + self.ROW_HALO)
        cost = ops.warp_cost(ref_map[:, ea:eb].contiguous(), src_maps, cams, hyp[:, :, ea:eb].contiguous(), geom=L.GEOM_PROJ,
                             cost=L.COST_VARIANCE_CVP, out_dtype=self.storage_dtype, ref_y0=ea)
        logits = self.cost_reg_refine(cost)
        o = ops.softargmin(logits, hyp[:, :, ea:eb].contiguous(), want_conf=want_conf, conf_mode=0)
        maps = [o["depth"]] + ([o["conf"]] if want_conf else [])
        mine = torch.stack([m[:, ra - ea:rb - ea].to(torch.float32) for m in maps], dim=1)        # [B, 1 or 2, owned rows, W]
        rows_max = max(b - a for a, b in bounds)
        full = pdist.gather_rows(mine, rows_max * world, rows_max, grp)                              # equal blocks, tail-padded
        full = torch.cat([full[:, :, r * rows_max:r * rows_max + (b - a)] for r, (a, b) in enumerate(bounds)], dim=2)
        return full[:, 0].contiguous(), (full[:, 1].contiguous() if want_conf else None)

    def _row_shard_applies(self, H: int) -> bool:
        """Decided identically on every rank (from H and the group size only), before any collective: every rank must own at least
        one 4-row block; smaller levels run replicated."""
        if self.row_group is None or H % 4:
            return False
        import torch.distributed as dist
        return H // 4 >= dist.get_world_size(self.row_group)

    def forward_train(self, ref_img, src_imgs, ref_in, src_in, ref_ex, src_ex, depth_min, depth_max, nscale):
        """train()-mode forward with autograd (reference net.py:96-229 with ``self.training``): 48 coarse planes, fixed
        halving refinement intervals, batch-statistics BatchNorm, the regulariser applied once per pyramid level.  The
        pyramid tower stays on PyTorch-ROCm autograd (upstream of the path); warp + cost and the regulariser + regression are
        the engine's autograd nodes.  The sampling grid carries no gradient (modules.py:83,241), so a refinement level's
        depth depends on the upsampled coarse depth only through ``sum_d p_d (depth_up + off_d) = depth_up + sum_d p_d off_d``:
        the engine regresses the per-batch offsets and the ``depth_up +`` stays an autograd add."""
        nsrc = len(src_imgs)
        dt = self.train_storage_dtype
        if self.feature_engine_train == "pscv":
            # the tower of ALL views in one engine pass (training.FeaturePyramidFn: no BatchNorm, the views are batch items);
            # channels-last 16-bit maps, which WarpCostFn takes as they are
            Bn = ref_img.shape[0]
            fdt = self.feature_engine_train_dtype or torch.float16      # the tower's own 16-bit format (fp16: see MVSNet.forward), then the sweep's
            levels = T.FeaturePyramidFn.apply(self.featurePyramid, fdt, nscale, torch.cat([ref_img] + list(src_imgs), 0),
                                              *T.FeaturePyramidFn.params(self.featurePyramid))
            if fdt != dt:
                levels = [lv.to(dt) for lv in levels]
            per_view = [torch.split(lv, Bn, 0) for lv in levels]                       # [level][view]
            ref_pyr = [pv[0] for pv in per_view]
            src_pyrs = [[pv[1 + i] for pv in per_view] for i in range(nsrc)]
            nchw = lambda f: (f.shape[0], f.shape[3], f.shape[1], f.shape[2])        # conditionIntrinsics reads [B,C,H,W] shapes
        else:
            ref_pyr = self.featurePyramid(ref_img, nscale)
            src_pyrs = [self.featurePyramid(s, nscale) for s in src_imgs]
            nchw = lambda f: tuple(f.shape)
        ref_in_ms = conditionIntrinsics(ref_in, ref_img.shape, [nchw(f) for f in ref_pyr])
        src_in_ms = torch.stack([conditionIntrinsics(src_in[:, i], ref_img.shape, [nchw(f) for f in src_pyrs[i]])
                                 for i in range(nsrc)]).permute(1, 0, 2, 3, 4)
        blocks = self.cost_reg_refine.train_blocks()
        params = T.RegressFn.block_params(blocks)

        def level_cost(level, hypos):
            cams = _cams(ref_in_ms[:, level], [src_in_ms[:, i, level] for i in range(nsrc)], ref_ex, [src_ex[:, i] for i in range(nsrc)])
            return T.WarpCostFn.apply(cams, hypos, L.GEOM_PROJ, L.COST_VARIANCE_CVP, dt, None, ref_pyr[level],
                                      *[p[level] for p in src_pyrs])

        hypos = calSweepingDepthHypo(ref_in_ms[:, -1], src_in_ms[:, 0, -1], ref_ex, src_ex, depth_min, depth_max,
                                     nhypothesis_init=48).to(torch.float32).contiguous()
        depth, conf = T.RegressFn.apply(blocks, hypos, dt, level_cost(nscale - 1, hypos), *params)
        depth_est_list = [depth]
        for id_level, level in enumerate(range(nscale - 2, -1, -1)):
            depth_up = F.interpolate(depth[None, :], size=None, scale_factor=2, mode='bicubic', align_corners=None).squeeze(0)
            interval = ((depth_max - depth_min) / 48 / 2 ** (id_level + 1)).to(torch.float32)              # net.py:178-181
            offs = torch.stack([i * interval for i in range(-4, 4)], dim=1).contiguous()                    # [B,8]
            hyp = (depth_up.detach().unsqueeze(1) + offs.view(-1, 8, 1, 1)).contiguous()                    # [B,8,H,W] for the warp
            resid, conf = T.RegressFn.apply(blocks, offs, dt, level_cost(level, hyp), *params)
            depth = depth_up + resid
            depth_est_list.append(depth)
        depth_est_list.reverse()
        return {"depth_est_list": depth_est_list, "prob_confidence": conf}

    def forward(self, ref_img, src_imgs, ref_in, src_in, ref_ex, src_ex, depth_min, depth_max, **kwargs):
        if self.training:
            return self.forward_train(ref_img, src_imgs, ref_in, src_in, ref_ex, src_ex, depth_min, depth_max,
                                      kwargs.get("nscale", self.nscale))
        nscale = kwargs.get("nscale", self.nscale)
        taps = kwargs.get("taps")
        nsrc = len(src_imgs)
        dt = self.storage_dtype
        engine = self.feature_engine == "pscv"
        with torch.no_grad():
            pyramid = (lambda x: self.featurePyramid.forward_engine(x, nscale, dt)) if engine else (lambda x: self.featurePyramid(x, nscale))
            if all(s.shape == ref_img.shape for s in src_imgs):
                # all views through the pyramid tower as one batch (same result as the per-view loop)
                levels = [torch.chunk(f, nsrc + 1, 0) for f in pyramid(ops.batch_views([ref_img] + list(src_imgs)))]
                ref_pyr = [lv[0] for lv in levels]
                src_pyrs = [[lv[i + 1] for lv in levels] for i in range(nsrc)]
            else:
                ref_pyr = pyramid(ref_img)
                src_pyrs = [pyramid(s) for s in src_imgs]
            # NCHW-style shapes of the levels (the engine's maps are [B,h,w,16])
            shp = (lambda f: (f.shape[0], f.shape[3], f.shape[1], f.shape[2])) if engine else (lambda f: tuple(f.shape))
            cl = (lambda f: f.contiguous()) if engine else (lambda f: ops.to_channels_last(f, dt))
            # Camera algebra: when all views share the image size, every per-level block (conditioned intrinsics, projection
            # stacks, the constants of calDepthHypo) comes from ONE launch (ops.cvp_cams); the tensor-level functions of
            # modules.py remain for direct callers and for views of different sizes.
            fast_cams = all(s.shape == ref_img.shape for s in src_imgs)
            if fast_cams:
                warp_cams, hypo_cams = ops.cvp_cams(ref_in, src_in, ref_ex, src_ex, [ref_img.shape[2] / shp(f)[2] for f in ref_pyr])
                fallback = ((depth_max - depth_min) / 128).to(torch.float32).reshape(-1).contiguous()
                ref_in_ms = src_in_ms = None
            else:
                ref_in_ms = conditionIntrinsics(ref_in, ref_img.shape, [shp(f) for f in ref_pyr])
                src_in_ms = torch.stack([conditionIntrinsics(src_in[:, i], ref_img.shape, [shp(f) for f in src_pyrs[i]])
                                         for i in range(nsrc)]).permute(1, 0, 2, 3, 4)

            # coarsest level: fronto-parallel sweep, 96 planes in eval mode (net.py:126-127)
            hypos = calSweepingDepthHypo(None if fast_cams else ref_in_ms[:, -1], None if fast_cams else src_in_ms[:, 0, -1], ref_ex, src_ex,
                                         depth_min, depth_max, nhypothesis_init=int(self.coarse_planes_eval)).to(torch.float32).contiguous()
            cams = warp_cams[-1] if fast_cams else _cams(ref_in_ms[:, -1], [src_in_ms[:, i, -1] for i in range(nsrc)], ref_ex,
                                                         [src_ex[:, i] for i in range(nsrc)])
            cost = ops.warp_cost(cl(ref_pyr[-1]), [cl(p[-1]) for p in src_pyrs],
                                 cams, hypos, geom=L.GEOM_PROJ, cost=L.COST_VARIANCE_CVP, out_dtype=dt)
            lt = {} if taps is not None else None
            logits = self.cost_reg_refine(cost, lt)
            is_last = nscale == 1
            o = ops.softargmin(logits, hypos, want_conf=is_last, conf_mode=0)
            depth = o["depth"]
            depth_est_list = [depth]
            if taps is not None:
                lt.update(cost=cost, logits=logits, hypos=hypos)
                taps.update(coarse=lt, refine=[])

            for id_level, level in enumerate(range(nscale - 2, -1, -1)):
                depth_up = F.interpolate(depth[None, :], size=None, scale_factor=2, mode='bicubic', align_corners=None).squeeze(0)
                if fast_cams and taps is None and self._row_shard_applies(depth_up.shape[-2]):
                    hyp = ops.cvp_depth_hypos(depth_up.to(torch.float32).contiguous(), hypo_cams[level], fallback)
                    depth, conf = self._refine_level_row_shard(cl(ref_pyr[level]), [cl(p[level]) for p in src_pyrs], warp_cams[level], hyp,
                                                               want_conf=level == 0)
                    o = {"depth": depth, "conf": conf}
                    depth_est_list.append(depth)
                    continue
                if fast_cams:
                    hyp = ops.cvp_depth_hypos(depth_up.to(torch.float32).contiguous(), hypo_cams[level], fallback)
                    cost = ops.warp_cost(cl(ref_pyr[level]), [cl(p[level]) for p in src_pyrs], warp_cams[level], hyp,
                                         geom=L.GEOM_PROJ, cost=L.COST_VARIANCE_CVP, out_dtype=dt)
                else:
                    hyp = calDepthHypo(depth_up, ref_in_ms[:, level], src_in_ms[:, :, level], ref_ex, src_ex, depth_min,
                                       depth_max, level).contiguous()
                    cost = proj_cost(nsrc, ref_pyr[level], src_pyrs, level, ref_in_ms[:, level], src_in_ms[:, :, level],
                                     ref_ex, src_ex, hyp, storage_dtype=dt, channels_last=engine)
                lt = {} if taps is not None else None
                logits = self.cost_reg_refine(cost, lt)
                is_last = level == 0
                o = ops.softargmin(logits, hyp, want_conf=is_last, conf_mode=0)
                depth = o["depth"]
                depth_est_list.append(depth)
                if taps is not None:
                    lt.update(cost=cost, logits=logits, hypos=hyp)
                    taps["refine"].append(lt)
        depth_est_list.reverse()
        return {"depth_est_list": depth_est_list, "prob_confidence": o["conf"]}
