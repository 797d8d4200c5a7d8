"""Drop-in for the live parts of the reference's ``models/VisMVSNet/model_cas.py`` on the pscv engine.

Per cascade stage (reference model_cas.py:303-420, mode 'soft'):
    one fused launch  warp (HOMOG geometry) + group-wise correlation for ALL source views
    per source view   3-D U-Net `Reg` (7 MFMA conv launches, the cat() is a channel-slice view) -> 1-channel head
                      (dot2 kernel) -> fused softmax / expected index / entropy -> 2-D UncertNet (PyTorch-ROCm)
    one launch        visibility-weighted fusion of the pair volumes (pscv_fuse_pairs)
    `RegFuse`         U-Net + head -> fused softmax with the +-2 window probability
Same module tree and state-dict keys as the reference (FeatExt / Reg / RegPair / RegFuse / UncertNet / SingleStage /
Model), so released checkpoints load unchanged."""
from __future__ import annotations

from typing import Dict, Optional

import contextlib

import torch
import torch.nn as nn

from ... import _lib as L
from ... import ops
from ... import training as T
from .nn_utils import UNet

cpg = 8


class FeatExt(nn.Module):
    """Upstream 2-D extractor: three 32-channel maps at 1/8, 1/4, 1/2 (reference model_cas.py:18-35)."""

    def __init__(self):
        super().__init__()
        self.init_conv = nn.Sequential(nn.Conv2d(3, 16, 5, 2, 2, bias=False), nn.BatchNorm2d(16), nn.ReLU())
        self.unet = UNet(16, 2, 1, 2, [], [32, 64, 128], [], '2d', 2)
        self.final_conv_1 = nn.Conv2d(128, 32, 3, 1, 1, bias=False)
        self.final_conv_2 = nn.Conv2d(64, 32, 3, 1, 1, bias=False)
        self.final_conv_3 = nn.Conv2d(32, 32, 3, 1, 1, bias=False)

    def forward(self, x):
        o1, o2, o3 = self.unet(self.init_conv(x), multi_scale=3)
        return self.final_conv_1(o1), self.final_conv_2(o2), self.final_conv_3(o3)

    # -- HIP path: every layer of the 2-D residual U-Net on pscv_conv2d_ex, channels-last 16-bit maps ----------
    def engine_layers(self, dtype: torch.dtype):
        """Packed weights + folded eval-mode BatchNorm of all 37 layers (the two transposed convs as four parity
        sub-convolutions each), rebuilt when a parameter / buffer changes."""
        key = (ops.weights_epoch(), dtype) + tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))
        if getattr(self, "_lay", None) is not None and self._lay_key == key:
            return self._lay
        mk = ops.Conv2dLayer.build

        def block(b):
            d = dict(c1=mk(b.conv1.weight, stride=b.stride, bn=_bn_tuple(b.bn1), bn_eps=b.bn1.eps, relu=True, dtype=dtype),
                     c2=mk(b.conv2.weight, stride=1, bn=_bn_tuple(b.bn2), bn_eps=b.bn2.eps, relu=True, dtype=dtype))   # relu after the add
            if b.downsample is not None:
                d["ds"] = mk(b.downsample[0].weight, stride=b.stride, bn=_bn_tuple(b.downsample[1]), bn_eps=b.downsample[1].eps,
                             dtype=dtype)
            return d
        lay = dict(init=mk(self.init_conv[0].weight, stride=2, bn=_bn_tuple(self.init_conv[1]), bn_eps=self.init_conv[1].eps,
                           relu=True, dtype=dtype),
                   enc=[[block(b) for b in seq] for seq in self.unet.enc_blocks.values()], dec=[])
        for parts in self.unet.dec_blocks.values():
            lay["dec"].append(dict(up=[mk(w_, stride=1, dtype=dtype) for w_ in ops.deconv2d_parity_weights(parts[0].weight)],
                                   conv=mk(parts[1].weight, stride=1, dtype=dtype), blocks=[block(b) for b in parts[2]]))
        lay["final"] = [mk(c.weight, stride=1, dtype=dtype) for c in (self.final_conv_1, self.final_conv_2, self.final_conv_3)]
        self._lay, self._lay_key = lay, key
        return lay

    def forward_engine(self, x: torch.Tensor, dtype: torch.dtype):
        """[B,3,H,W] images on the GPU -> three channels-last feature maps [B,H/8,W/8,32], [B,H/4,W/4,32], [B,H/2,W/2,32] in
        ``dtype`` (eval mode): the layout the warp kernel reads."""
        if self.training:
            raise NotImplementedError("pscv FeatExt: the HIP path is inference-only (eval-mode BatchNorm is folded)")
        ly = self.engine_layers(dtype)

        def run_block(y, d):
            t = ops.conv2d(y, d["c1"])
            sc = ops.conv2d(y, d["ds"]) if "ds" in d else y
            return ops.conv2d(t, d["c2"], skip=sc)
        y = ops.conv2d(ops.image_to_channels_last8(x, dtype), ly["init"])
        skips = []
        for blocks in ly["enc"]:
            for d in blocks:
                y = run_block(y, d)
            skips.append(y)
        outs = [y]
        for i, d in enumerate(ly["dec"]):
            skip = skips[-2 - i]
            B, H, W, f = skip.shape
            cat = torch.empty((B, H, W, 2 * f), dtype=dtype, device=x.device)       # [deconv | skip] (nn_utils.py:269-271)
            for par, sub in enumerate(d["up"]):
                ops.conv2d(y, sub, out=cat, out_coff=0, parity=par)
            cat[..., f:] = skip
            y = ops.conv2d(cat, d["conv"])
            for bd in d["blocks"]:
                y = run_block(y, bd)
            outs.append(y)
        return tuple(ops.conv2d(o, l) for o, l in zip(outs[-3:], ly["final"]))


def _bn_tuple(bn):
    return (bn.weight, bn.bias, bn.running_mean, bn.running_var)


def _featext_forward_train(self, x: torch.Tensor, groups: int, dtype: torch.dtype):
    """FeatExt in train() mode on the engine (``feature_engine_train = "pscv"``): the same graph as ``forward`` (reference
    model_cas.py:18-35, nn_utils.py:123-171,194-278) built from the engine's autograd nodes -- ``training.Conv2dFn`` / ``Deconv2dFn`` /
    ``BnAct2dFn`` on channels-last 16-bit maps; shortcuts and the decoder's concat are torch autograd's -- for ALL views at once:
    x [groups * B, 3, H, W] view-major, every BatchNorm normalising each view with its own batch statistics (the reference's per-view
    calls, frontend.py:59-61).  Returns the three maps [groups * B, h, w, 32] (1/8, 1/4, 1/2) in ``dtype``."""
    C2, D2, BA = T.Conv2dFn.apply, T.Deconv2dFn.apply, T.BnAct2dFn.apply
    tag = [0]

    def conv(y, m, stride):
        tag[0] += 1
        return C2(self, tag[0], dtype, stride, y, m.weight)

    def bnact(y, bn, relu, skip=None):
        return BA(bn, groups, relu, y, skip, bn.weight, bn.bias)

    def block(y, b):
        t = bnact(conv(y, b.conv1, b.stride), b.bn1, "pre")
        sc = y if b.downsample is None else bnact(conv(y, b.downsample[0], b.stride), b.downsample[1], None)
        return bnact(conv(t, b.conv2, 1), b.bn2, "post", sc)

    y = bnact(conv(ops.image_to_channels_last8(x.detach(), dtype), self.init_conv[0], 2), self.init_conv[1], "pre")
    skips = []
    for seq in self.unet.enc_blocks.values():
        for b in seq:
            y = block(y, b)
        skips.append(y)
    outs = [y]
    for i, parts in enumerate(self.unet.dec_blocks.values()):
        tag[0] += 1
        up = D2(self, tag[0], dtype, y, parts[0].weight)
        y = conv(torch.cat([up, skips[-2 - i]], 3), parts[1], 1)                      # [deconv | skip] (nn_utils.py:269-271)
        if len(parts) == 3:
            for b in parts[2]:
                y = block(y, b)
        outs.append(y)
    return tuple(conv(o, c, 1) for o, c in zip(outs[-3:], (self.final_conv_1, self.final_conv_2, self.final_conv_3)))


FeatExt.forward_train = _featext_forward_train


class _RegUNet(nn.Module):
    """Holder + engine executor of ``UNet(8, 1, 0, 4, [], [8, 16], [], tag, dim=3)`` (nn_utils.py:194-278)."""

    def __init__(self, tag: str):
        super().__init__()
        self.unet = UNet(8, 1, 0, 4, [], [8, 16], [], tag, dim=3)
        self._lay, self._key = None, None

    def _layers(self, dtype) -> Dict[str, ops.Conv3dLayer]:
        key = (ops.weights_epoch(), dtype) + tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))
        if self._lay is None or key != self._key:
            enc = list(self.unet.enc_blocks.values())
            b0, b1 = enc[0][0], enc[1][0]
            dec = list(self.unet.dec_blocks.values())[0]
            dev = b0.conv1.weight.device
            mk = ops.Conv3dLayer.build
            # the strided 1x1x1 shortcut conv is the centre tap of a 3x3x3 stride-2 conv (in = 2*out + 1 - 1), so the block's
            # conv1 and its shortcut are ONE 8 -> 32 stride-2 launch over the same input brick: channels 0-15 = relu(bn1(conv1)),
            # channels 16-31 = bn_ds(shortcut) (per-channel ReLU floor: 0 / -inf); conv2 then reads [0,16) and adds [16,32)
            ds_w = torch.zeros(16, 8, 3, 3, 3, dtype=torch.float32, device=dev)
            ds_w[:, :, 1, 1, 1] = b1.downsample[0].weight.detach().float().view(16, 8)
            bn_ds = b1.downsample[1]
            if b1.bn1.eps != bn_ds.eps:
                raise ValueError("pscv Vis U-Net: BasicBlock.bn1 and the shortcut's BatchNorm must share eps")
            e1_w = torch.cat([b1.conv1.weight.detach().float(), ds_w], dim=0)
            e1_bn = tuple(torch.cat([p_.detach().float(), q_.detach().float()]) for p_, q_ in zip(_bn_tuple(b1.bn1), _bn_tuple(bn_ds)))
            e1_floor = torch.cat([torch.zeros(16), torch.full((16,), float("-inf"))])
            self._lay = {
                "e0c1": mk(b0.conv1.weight, kind=L.CONV_S1, device=dev, bn=_bn_tuple(b0.bn1), bn_eps=b0.bn1.eps, relu=True, dtype=dtype),
                "e0c2": mk(b0.conv2.weight, kind=L.CONV_S1, device=dev, bn=_bn_tuple(b0.bn2), bn_eps=b0.bn2.eps, relu_post=True, dtype=dtype),
                "e1c1ds": mk(e1_w, kind=L.CONV_S2, device=dev, bn=e1_bn, bn_eps=b1.bn1.eps, relu=True, floor=e1_floor, dtype=dtype),
                "e1c2": mk(b1.conv2.weight, kind=L.CONV_S1, device=dev, bn=_bn_tuple(b1.bn2), bn_eps=b1.bn2.eps, relu_post=True, dtype=dtype),
                "dec": mk(dec[0].weight, kind=L.CONV_T2, transposed=True, device=dev, dtype=dtype),
                "post": mk(dec[1].weight, kind=L.CONV_S1, device=dev, dtype=dtype),
            }
            self._key = key
        return self._lay

    # True: enc0's two convolutions as ONE depth sweep with the intermediate volume in LDS (pscv_conv3d_block8, same bits).  Built
    # and measured in round 3: 147-157 us against 126 us for the two launches at 16 x 576 x 800 (its fetch / MFMA / epilogue phases
    # run back to back inside a workgroup and the halo recompute costs 29 % more MFMAs), so the two launches stay the default.
    FUSED_BLOCK = False

    def run_unet(self, x: torch.Tensor) -> torch.Tensor:
        """x [n,d,h,w,8] 16-bit channels-last -> [n,d,h,w,8]."""
        if self.training:
            raise RuntimeError("pscv Vis U-Net: in train() mode the U-Net runs inside training.VisUNetFn (SingleStage.forward "
                               "routes there); this entry point is the eval-mode engine")
        n, d, h, w, _ = x.shape
        if d % 2 or h % 2 or w % 2:
            raise ValueError(f"Vis U-Net needs even d,h,w (got {d},{h},{w}), as in the reference")
        ly = self._layers(x.dtype)
        # the full-resolution BasicBlock (conv1 + bn + relu, conv2 + bn, + x, relu) as ONE depth sweep: its intermediate volume
        # stays in LDS (same values); layers that are not on the depth-sweep kernels run as two launches
        e0 = ops.conv3d_block8(x, ly["e0c1"], ly["e0c2"], residual=True) if self.FUSED_BLOCK else None
        if e0 is None:
            t = ops.conv3d(x, ly["e0c1"])
            e0 = ops.conv3d(t, ly["e0c2"], skip=x)
        t1ds = ops.conv3d(e0, ly["e1c1ds"])                                       # [relu(bn1(conv1)) | bn_ds(shortcut)]
        e1 = ops.conv3d(t1ds, ly["e1c2"], skip=t1ds, skip_coff=16)
        up = ops.conv3d(e1, ly["dec"])
        if ly["post"].kind == L.CONV_S1P8:
            # cat([deconv, enc0]) (nn_utils.py:269-271) is gathered by the decoder conv's staging: both producers store dense
            # 8-channel volumes (half-voxel stores into a 16-channel buffer cost them 20-50 us each at 256 x 144 x 200)
            return ops.conv3d(up, ly["post"], x2=e0)
        return ops.conv3d(torch.cat([up, e0], dim=4), ly["post"])


class Reg(_RegUNet):            # reference model_cas.py:38-48
    def __init__(self):
        super().__init__('reg1')
        self.init_conv = lambda x: x

    def forward(self, x):
        return self.run_unet(x)


class _Head(nn.Module):
    def __init__(self):
        super().__init__()
        self.final_conv = nn.Conv3d(8, 1, 3, 1, 1, bias=False)
        self._hl, self._hk = None, None

    def head_layer(self, dtype):
        w = self.final_conv.weight
        key = (ops.weights_epoch(), dtype, w.data_ptr(), w._version)
        if self._hl is None or key != self._hk:
            self._hl, self._hk = ops.Conv3dLayer.build(w, kind=L.CONV_S1, device=w.device, dtype=dtype), key
        return self._hl

    def head(self, x: torch.Tensor) -> torch.Tensor:
        """[n,d,h,w,8] -> fp32 scores [n,d,h,w]."""
        return ops.conv3d(x, self.head_layer(x.dtype), out_dtype=torch.float32).squeeze(-1)


class RegPair(_Head):           # reference model_cas.py:51-59
    def forward(self, x):
        return self.head(x)

    def head_index_entropy(self, x, index, entropy, want_scores=False):
        """The head fused with soft_argmin + entropy (``ops.head_index_entropy``); None when this size takes the two-launch path."""
        self.head_layer(x.dtype)
        return ops.head_index_entropy(x, self._hl, index, entropy, want_scores=want_scores)


class RegFuse(_RegUNet):        # reference model_cas.py:62-74
    def __init__(self):
        super().__init__('reg2')
        self.init_conv = lambda x: x
        self.final_conv = nn.Conv3d(8, 1, 3, 1, 1, bias=False)
        self._hl, self._hk = None, None

    head = _Head.head
    head_layer = _Head.head_layer

    def forward(self, x):
        return self.head(self.run_unet(x))


class UncertNet(nn.Module):
    """2-D entropy -> log-uncertainty net (reference model_cas.py:77-98).  Eval mode on the GPU: one
    fused HIP launch (``pscv_uncert_net``); train mode / autograd: the torch layers (batch-statistics BatchNorm)."""

    def __init__(self, num_heads=1):
        super().__init__()
        self.conv1 = nn.Sequential(nn.Conv2d(1, 8, 3, 1, 1, bias=False), nn.BatchNorm2d(8), nn.ReLU())
        self.conv2 = nn.Sequential(nn.Conv2d(8, 8, 3, 1, 1, bias=False), nn.BatchNorm2d(8), nn.ReLU())
        self.head_convs = nn.ModuleList([nn.Conv2d(8, 1, 3, 1, 1, bias=False) for _ in range(num_heads)])

    def engine_params(self) -> torch.Tensor:
        """The folded parameter block of ``ops.uncert_net``, rebuilt when a parameter / buffer changes."""
        ts = list(self.parameters()) + list(self.buffers())
        key = (ops.weights_epoch(),) + tuple((t.data_ptr(), t._version) for t in ts)
        if getattr(self, "_prm", None) is None or self._prm_key != key:
            c1, n1, c2, n2 = self.conv1[0], self.conv1[1], self.conv2[0], self.conv2[1]
            self._prm = ops.pack_uncert_params(c1.weight, _bn_tuple(n1), c2.weight, _bn_tuple(n2), self.head_convs[0].weight, n1.eps, n2.eps)
            self._prm_key = key
        return self._prm

    def forward(self, x):
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        if not self.training and not needs_grad and x.is_cuda and x.dtype == torch.float32 and len(self.head_convs) == 1:
            # eval hot path: the three convolutions, two BatchNorms and the residual add in ONE launch (csrc/uncert_net.hip).
            # Forward-only: a frozen (.eval()) UncertNet inside a training step -- SingleStage.forward_train feeds it an entropy
            # that requires grad -- takes the differentiable branch below, or the gradient to the entropy / pair branch is lost.
            n, _, h, w = x.shape
            return [ops.uncert_net(x.reshape(n, h, w).contiguous(), self.engine_params()).view(n, 1, h, w)]
        out = self.conv2(self.conv1(x))
        out = out + x
        return [conv(out) for conv in self.head_convs]


class SingleStage(nn.Module):
    def __init__(self):
        super().__init__()
        self.reg = Reg()
        self.reg_fuse = RegFuse()
        self.reg_pair = RegPair()
        self.uncert_net = UncertNet(1)
        self.storage_dtype = torch.float16
        self.train_storage_dtype = torch.bfloat16   # train(): bf16 activations / gradients by default (range), fp32 accumulation
        # source-view shard: with a torch.distributed group set here, rank r warps and
        # regularises source views r, r+G, ... only; the visibility-weighted sums are reduce-scattered (RCCL, 16-bit shares)
        # into per-rank depth / row slabs and RegFuse runs slab-sharded (``_fuse_view_shard``).  The reference has no
        # counterpart (it loops over all views).
        self.view_group = None
        # source-view shard: True (default) = the fused volume is reduce-scattered into per-rank slabs and RegFuse runs on the
        # slab (+ recomputed halo); False = 16-bit all-reduce + replicated RegFuse at every stage (measurement / tests)
        self.view_slabs = True
        self.view_reduce_fp32 = False   # source-view shard: reduce fp32 shares of the fused volume instead of 16-bit ones (2x the payload)
        # depth-plane shard: with a group set here, rank r sweeps, regularises and fuses only
        # the planes it owns plus a 16-plane halo per side (the pair U-Net + head and the fuse U-Net + head each reach 8
        # planes), and the softmax over D is merged from per-rank partials.  The reference has no counterpart.
        self.depth_group = None
        # row-slab shard (round 4; the cascade's stages 2-3 have 32 / 16 planes per-pixel: nothing to shard along depth): with a
        # group set here, rank r runs the WHOLE stage -- warp, pair U-Nets, UncertNet, fusion, fuse U-Net, heads -- on the image rows
        # it owns plus a recomputed 16-row halo per side (pair U-Net + head and fuse U-Net + head reach 8 rows each), with the
        # reference feature map, the per-pixel depth starts and the reference principal point cropped to the slab; the only
        # communication is ONE all-gather of the owned rows of the stage's small output maps.  The reference has no counterpart.
        self.row_group = None
        # pair branch: RegPair's head, soft_argmin and the entropy as one launch (pscv_head_index_entropy); False = the head
        # and pscv_softargmin as two launches with the fp32 score volume in between (the entropy then keeps the reference's clamp)
        self.fused_pair_head = True

    def build_cost_volume(self, ref, ref_cam, srcs, srcs_cam, depth_num, depth_start, depth_interval, s_scale, ref_y0: int = 0):
        """Pair-wise group-correlation volumes of ALL source views in one fused launch: [n_src,n,d,h,w,8]
        (reference model_cas.py:176-186 + groupwise_correlation at :340)."""
        cl = getattr(self, "_channels_last_features", False)     # the HIP extractor already wrote [n,h,w,32] 16-bit maps
        (n, h, w, _) = ref.shape if cl else (ref.shape[0], ref.shape[2], ref.shape[3], 0)
        steps = torch.arange(depth_num, dtype=torch.float32, device=ref.device).view(1, depth_num, 1, 1)
        planes = (depth_start + depth_interval * steps).to(torch.float32)            # homography.py:39-41
        planes = planes.reshape(n, depth_num) if planes.shape[2:] == (1, 1) else planes.expand(n, depth_num, h, w)
        cams = ops.homog_cams_device(ref_cam, srcs_cam, 1.0 / s_scale)
        ref_cl = ref.contiguous() if cl else ops.to_channels_last(ref, self.storage_dtype)
        srcs_cl = [s.contiguous() if cl else ops.to_channels_last(s, self.storage_dtype) for s in srcs]
        return ops.warp_cost(ref_cl, srcs_cl, cams, planes.contiguous(), geom=L.GEOM_HOMOG, cost=L.COST_GROUPCORR,
                             out_dtype=self.storage_dtype, ref_y0=ref_y0)

    def forward_train(self, ref_feat, ref_cam, srcs_feat, srcs_cam, depth_num, depth_start, depth_interval, s_scale):
        """One cascade stage in train() mode with autograd (reference model_cas.py:303-420 under ``loss.backward()``): the
        fused warp + group correlation, the pair / fuse U-Nets (batch-statistics BatchNorm), the score heads with expected
        index + entropy and the visibility-weighted fusion are the engine's autograd nodes (training.WarpCostFn / VisUNetFn /
        ScoreHeadFn / FusePairsFn); the 2-D ``UncertNet`` on the entropy map stays on PyTorch-ROCm autograd.  The homographies
        carry no gradient (homography.py:25,92,110)."""
        dt = self.train_storage_dtype
        if ref_feat.dtype == dt and dt != torch.float32:          # channels-last 16-bit maps from the engine's extractor (FeatExt.forward_train)
            n, h, w, _ = ref_feat.shape
        else:
            n, _, h, w = ref_feat.shape
        if depth_num % 2 or h % 2 or w % 2:
            raise ValueError(f"Vis U-Net needs even d,h,w (got {depth_num},{h},{w}), as in the reference")
        steps = torch.arange(depth_num, dtype=torch.float32, device=ref_feat.device).view(1, depth_num, 1, 1)
        planes = (depth_start.detach() + depth_interval.detach() * steps).to(torch.float32)
        planes = planes.reshape(n, depth_num) if planes.shape[2:] == (1, 1) else planes.expand(n, depth_num, h, w)
        cams = ops.homog_cams_device(ref_cam.detach(), [c.detach() for c in srcs_cam], 1.0 / s_scale)
        costs = T.WarpCostFn.apply(cams, planes.contiguous(), L.GEOM_HOMOG, L.COST_GROUPCORR, dt, None, ref_feat, *srcs_feat)
        reg_params = T.VisUNetFn.params(self.reg)
        interms, uncerts, pair_results = [], [], []
        # the pair branch of ALL source views in one pass per layer: the views are the group axis of the BatchNorm statistics (each
        # view its own, like the reference's per-view calls), convolutions / weight gradients / the score head run over the stack
        V = len(srcs_feat)
        interm_all = T.VisUNetFn.apply(self.reg, dt, V, costs.reshape((V * n,) + tuple(costs.shape[2:])), *reg_params)
        idx_all, ent_all, _ = T.ScoreHeadFn.apply(dt, None, interm_all, self.reg_pair.final_conv.weight)
        per_view = lambda t_: t_.view((V, n) + tuple(t_.shape[1:])).unbind(0)         # (one stack in the backward, not V padded adds)
        for interm, idx, ent in zip(per_view(interm_all), per_view(idx_all), per_view(ent_all)):
            est_depth = idx.unsqueeze(1) * depth_interval + depth_start                  # model_cas.py:348
            heads = self.uncert_net(ent.unsqueeze(1))
            pair_results.append([est_depth, heads])
            interms.append(interm)
            uncerts.append(heads[0].squeeze(1))
        fused = T.FusePairsFn.apply(len(interms), *interms, *uncerts)                    # model_cas.py:354-357,385-386
        fu = T.VisUNetFn.apply(self.reg_fuse, dt, 1, fused, *T.VisUNetFn.params(self.reg_fuse))
        idx, _, conf = T.ScoreHeadFn.apply(dt, 2.0, fu, self.reg_fuse.final_conv.weight)
        est_depth = idx.unsqueeze(1) * depth_interval + depth_start                      # model_cas.py:404-405
        return est_depth, conf.unsqueeze(1), pair_results

    @staticmethod
    def _merged_head(score_ext, a, b, ea, grp, window=None):
        """Softmax statistics over ALL depth planes from a rank's scores on [ea, ea + score_ext.shape[1]): log-sum-exp partials
        (max, sum e, sum e*logit, sum e*index) of the OWNED planes [a, b) (none: a neutral partial), one small all-gather, local
        merge -> (expected index, entropy, +-window probability or None); the window probability is one more all-reduce of a
        [n,h,w] map."""
        import torch.distributed as dist
        world = dist.get_world_size(grp)
        n, _, h, w = score_ext.shape
        if b > a:
            own = score_ext[:, a - ea:b - ea].contiguous()
            part = ops.softargmin(own, own, want_partials=True, index_offset=a)["partials"]     # "depth" := the logits
        else:
            part = torch.zeros((n, 4, h, w), dtype=torch.float32, device=score_ext.device)
            part[:, 0] = -3.0e38
        bufs = [torch.empty_like(part) for _ in range(world)]
        dist.all_gather(bufs, part, group=grp)
        st = torch.stack(bufs)                                                          # [world,n,4,h,w]
        m = st[:, :, 0].max(dim=0).values
        f = torch.exp(st[:, :, 0] - m.unsqueeze(0))
        Z, sl, si = (st[:, :, 1] * f).sum(0), (st[:, :, 2] * f).sum(0), (st[:, :, 3] * f).sum(0)
        idx = si / Z
        entropy = m + torch.log(Z) - sl / Z               # -sum p log p = log-sum-exp - E[logit]
        conf = None
        if window is not None:
            if b > a:
                conf = ops.softargmin_window(own, torch.stack([m, Z, idx], dim=1).contiguous(), window=window, index_offset=a)
            else:
                conf = torch.zeros((n, h, w), dtype=torch.float32, device=score_ext.device)
            dist.all_reduce(conf, group=grp)
        return idx, entropy, conf

    def _fuse_view_shard(self, interms, uncerts, grp):
        """Source-view shard, fusion + fuse net (reference model_cas.py:354-357,385-405 across ranks).  Rank r holds the pair
        volumes of ITS source views.  (1) all-reduce of the weight sums [n,h,w] (tiny); (2) each rank normalises its partial
        sum by the TOTAL weight and rounds it to the storage format: an additive 16-bit share of the fused volume; (3)
        ``dist.reduce_to_slab``: reduce-scatter of the shares into per-rank slabs along depth or rows (whichever slab is
        thicker) + 8-unit halos from the two neighbours; (4) RegFuse + head on the slab (the halo is recomputed: values more
        than 8 units inside an artificial border equal the unsharded ones); (5) heads: depth slabs merge log-sum-exp partials,
        row slabs regress locally and all-gather their rows.  Stages whose slabs would be thinner than the halo reduce the
        16-bit shares with one all-reduce and run RegFuse replicated.  Returns (expected index [n,h,w], +-2 probability [n,h,w],
        fused volume or None)."""
        import torch.distributed as dist
        from ... import dist as pdist
        world = dist.get_world_size(grp)
        part, wsum = ops.fuse_pairs(interms, uncerts, normalise=False, want_wsum=True)
        dist.all_reduce(wsum, group=grp)                                               # sum_v w_v over ALL views
        store = interms[0].dtype
        if self.view_reduce_fp32:
            # fp32 shares: the cross-rank sum is ONE rounding to the storage format (after the reduction) instead of one per rank
            # plus RCCL's 16-bit partial sums in ring / tree order -- twice the payload (round 5: tests/test_gpu_dist.py prints the
            # 8-rank depth error with and without; 16-bit shares are the default, the budget SURVEY 8e counts)
            share = part / wsum.view(wsum.shape[0], 1, wsum.shape[1], wsum.shape[2], 1)
        else:
            share = ops.fuse_finish(part, wsum, store)                                 # (sum_{v in rank} w_v interm_v) / sum_v w_v
        n, d, h, w, _ = share.shape
        plan = pdist.slab_axis(d, h, world) if self.view_slabs else None
        if plan is None:
            dist.all_reduce(share, group=grp)                                          # 16-bit (or fp32) payload, replicated fuse net
            share = share.to(store)
            score = self.reg_fuse(share)
            o = ops.softargmin(score, None, want_index=True, want_conf=True, conf_mode=1, window=2.0)
            return o["index"], o["conf"], share, score
        axis, S = plan
        ext, lo, a, b = pdist.reduce_to_slab(share, axis, grp)
        score = self.reg_fuse(ext.to(store)) if ext is not None else None              # fp32 scores on the extended slab
        if axis == 1:
            if score is None:
                score = torch.zeros((n, 0, h, w), dtype=torch.float32, device=share.device)
            idx, _, conf = self._merged_head(score, a, b, lo, grp, window=2.0)
            return idx, conf, None, None
        if score is not None:
            o = ops.softargmin(score[:, :, a - lo:b - lo].contiguous(), None, want_index=True, want_conf=True, conf_mode=1, window=2.0)
            rows = torch.stack([o["index"], o["conf"]], dim=1)                         # [n,2,valid,w]
        else:
            rows = torch.zeros((n, 2, 0, w), dtype=torch.float32, device=share.device)
        full = pdist.gather_rows(rows, h, S, grp)
        return full[:, 0], full[:, 1], None, None

    PAIR_BATCH_BYTES = 48 << 20   # pair volumes (8 channels, 16-bit) batched into one U-Net pass: at most this many bytes per group
    # Groups of pair passes on HIP streams (forked from / joined into the caller's stream): an EXPERIMENT, off (1).  Measured
    # (scripts/dev/vis_pair_streams.py, rounds 4 and 6): two streams 15.99 -> 15.68 ms at configuration 5, 2.320 -> 2.296 ms at
    # configuration 3, more streams no better (ROCm 7.2 keeps two branches of a captured graph in flight).  Not the default: the pair
    # volumes are ALLOCATED on the side streams and read on the caller's stream after the join -- the caching allocator may hand such a
    # block to a later side-stream allocation while the caller's stream still reads it; round 6 switched it on, the unsharded forward
    # stayed bit-identical, the two-rank source-view shard at configuration 5 did not (depth 6.8e-3 off:
    # tests/test_gpu_fullsize.py::test_vis_fullsize_shard_equals_unsharded).  A safe version allocates every cross-stream tensor on the
    # caller's stream first, as CVP's FeaturePyramid.forward_engine_split does.
    PAIR_STREAMS = 1
    DEPTH_HALO = 16   # planes of redundant compute per side: 8 (pair U-Net + head) + 8 (fuse U-Net + head)

    def forward_depth_shard(self, ref_feat, ref_cam, srcs_feat, srcs_cam, depth_num, depth_start, depth_interval, s_scale):
        """Eval-mode stage with the depth planes sharded over ``self.depth_group``.  Per rank: warp + group correlation,
        pair U-Nets, fusion and fuse U-Net on the owned planes [a, b) extended by the halo (no communication: the halo is
        recomputed, and values more than 8 planes inside an artificial boundary equal the unsharded ones exactly); the
        softmax statistics come from log-sum-exp partials of the OWNED planes -- (max, sum e, sum e*logit, sum e*index): one
        small all-gather per head -- and the +-2 window probability from one more all-reduce of a [n,h,w] map."""
        import torch.distributed as dist
        from ... import dist as pdist
        grp = self.depth_group
        world, rank = dist.get_world_size(grp), dist.get_rank(grp)
        a, b = pdist.plane_shard(depth_num, world, rank, multiple=2)
        ea, eb = max(0, a - self.DEPTH_HALO), min(depth_num, b + self.DEPTH_HALO)
        costs = self.build_cost_volume(ref_feat, ref_cam, srcs_feat, srcs_cam, eb - ea, depth_start + depth_interval * ea,
                                       depth_interval, s_scale)

        merged = lambda score_ext, window=None: self._merged_head(score_ext, a, b, ea, grp, window)

        interms, uncerts, pair_results = [], [], []
        for i in range(len(srcs_feat)):
            interm = self.reg(costs[i])
            idx, ent, _ = merged(self.reg_pair(interm))
            est_depth = idx.unsqueeze(1) * depth_interval + depth_start
            heads = self.uncert_net(ent.unsqueeze(1))
            pair_results.append([est_depth, heads])
            interms.append(interm)
            uncerts.append(heads[0].squeeze(1).to(torch.float32).contiguous())
        fused = ops.fuse_pairs(interms, uncerts)
        idx, _, conf = merged(self.reg_fuse(fused), window=2.0)
        est_depth = idx.unsqueeze(1) * depth_interval + depth_start
        return est_depth, conf.unsqueeze(1), pair_results

    # rows a stage's outputs depend on beyond their own: pair U-Net + head 8, the 2-D UncertNet between the pair branch and the fusion
    # (three 3x3 convolutions over (h, w)) 3, fuse U-Net + head 8 = 19; rounded up to a multiple of 4 so that the stride-2 level of
    # the U-Nets sees the slab in the phase it has in the image (round 5: was 16 -- the UncertNet rows were missing, so the 3 owned
    # rows next to an interior slab boundary saw entropy rows contaminated by the artificial zero border)
    ROW_HALO = 20

    def forward_row_shard(self, ref_feat, ref_cam, srcs_feat, srcs_cam, depth_num, depth_start, depth_interval, s_scale):
        """Eval-mode stage with the IMAGE ROWS sharded over ``self.row_group`` (every plane on every rank).  Rank r owns rows
        [ra, rb) (boundaries multiples of 4: the U-Net's stride-2 level sees the slab in the phase it has in the image) and runs the
        unsharded stage on rows [ra - ROW_HALO, rb + ROW_HALO) clipped to the image: the reference feature map and the per-pixel depth starts are
        cropped, the cameras stay those of the whole image and the warp evaluates slab row y at (x, y + slab origin)
        (`pscv_warp_cost_rows`: the cost volume of the slab is bit-identical to those rows of the unsharded one; the source maps stay
        whole).  Values ROW_HALO (>= 19) rows or more inside an artificial border therefore equal the unsharded ones; rows at the image border
        keep their zero padding.  One all-gather per stage of the owned rows of (depth, probability, pair depths, pair
        uncertainties): (2 + 2 n_src) maps of h x w floats in all."""
        import torch.distributed as dist
        from ... import dist as pdist
        grp = self.row_group
        world, rank = dist.get_world_size(grp), dist.get_rank(grp)
        cl = getattr(self, "_channels_last_features", False)
        h = ref_feat.shape[1] if cl else ref_feat.shape[2]
        if h % 4:
            raise ValueError(f"row shard: the stage height {h} must be a multiple of 4")
        bounds = [pdist.plane_shard(h, world, r, multiple=4) for r in range(world)]
        ra, rb = bounds[rank]
        empty = [r for r, (a, b) in enumerate(bounds) if b <= a]
        if empty:      # decided identically on EVERY rank, before any collective (a raise on the empty ranks only would wedge the others)
            raise ValueError(f"row shard: {world} ranks for {h} rows leaves rank(s) {empty} without a 4-row block")
        ea, eb = max(0, ra - self.ROW_HALO), min(h, rb + self.ROW_HALO)
        ref_slab = (ref_feat[:, ea:eb] if cl else ref_feat[:, :, ea:eb]).contiguous()
        start = depth_start if depth_start.shape[-2] == 1 else depth_start[:, :, ea:eb].contiguous()
        self._row_y0 = ea               # the warp evaluates slab row y at (x, y + ea) with the WHOLE image's cameras: same bits as unsharded
        try:
            est, prob, pairs = self.forward((ref_slab, ref_cam, srcs_feat, srcs_cam), depth_num, depth_start_override=start,
                                            depth_interval_override=depth_interval, s_scale=s_scale)
        finally:
            self._row_y0 = None
        maps = [est, prob] + [m for ed, hd in pairs for m in (ed, hd[0])]                    # each [n,1,slab rows,w]
        mine = torch.cat([m[:, :, ra - ea:rb - ea].to(torch.float32) for m in maps], dim=1)  # [n, 2 + 2 n_src, owned rows, w]
        rows_max = max(b - a for a, b in bounds)
        padded = mine.new_zeros(mine.shape[:2] + (rows_max, mine.shape[3]))
        padded[:, :, :rb - ra] = mine
        gathered = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(gathered, padded.contiguous(), group=grp)
        full = torch.cat([g[:, :, :b - a] for g, (a, b) in zip(gathered, bounds)], dim=2)   # [n, 2 + 2 n_src, h, w]
        est_full, prob_full = full[:, 0:1], full[:, 1:2]
        pair_results = [[full[:, 2 + 2 * i:3 + 2 * i], [full[:, 3 + 2 * i:4 + 2 * i]]] for i in range(len(pairs))]
        return est_full, prob_full, pair_results

    def forward(self, sample, depth_num, upsample=False, mem=False, mode='soft', depth_start_override=None,
                depth_interval_override=None, s_scale=1, taps: Optional[dict] = None):
        if mem or mode != 'soft' or upsample:
            raise NotImplementedError("pscv Vis-MVSNet implements the reference's live configuration: mode='soft', "
                                      "no mem, no upsample (frontend.py:28-30)")
        ref_feat, ref_cam, srcs_feat, srcs_cam = sample
        depth_start = ref_cam[:, 1:2, 3:4, 0:1] if depth_start_override is None else depth_start_override
        depth_interval = ref_cam[:, 1:2, 3:4, 1:2] if depth_interval_override is None else depth_interval_override
        if self.training:
            return self.forward_train(ref_feat, ref_cam, srcs_feat, srcs_cam, depth_num, depth_start, depth_interval, s_scale)
        if self.row_group is not None and getattr(self, "_row_y0", None) is None:
            if self.view_group is not None or self.depth_group is not None:
                raise NotImplementedError("pscv Vis-MVSNet: one shard per stage (rows, depth planes or source views)")
            return self.forward_row_shard(ref_feat, ref_cam, srcs_feat, srcs_cam, depth_num, depth_start, depth_interval, s_scale)
        if self.depth_group is not None:
            if self.view_group is not None:
                raise NotImplementedError("pscv Vis-MVSNet: choose the depth-plane shard or the source-view shard, not both")
            return self.forward_depth_shard(ref_feat, ref_cam, srcs_feat, srcs_cam, depth_num, depth_start, depth_interval, s_scale)
        n_views = len(srcs_feat)
        world, rank = 1, 0
        if self.view_group is not None:
            import torch.distributed as dist
            world, rank = dist.get_world_size(self.view_group), dist.get_rank(self.view_group)
        mine = [i for i in range(n_views) if i % world == rank]
        srcs_feat, srcs_cam = [srcs_feat[i] for i in mine], [srcs_cam[i] for i in mine]
        if not mine:
            raise ValueError("view shard: more ranks than source views")
        costs = self.build_cost_volume(ref_feat, ref_cam, srcs_feat, srcs_cam, depth_num, depth_start, depth_interval, s_scale,
                                       ref_y0=getattr(self, "_row_y0", None) or 0)
        interms, uncerts, pair_results = [], [], []
        # the pair branch of ALL source views as ONE batch per layer (the views share `reg` / `reg_pair`, and the fused warp
        # launch already wrote their volumes back to back: [n_src, n, d, h, w, 8] -> [n_src * n, d, h, w, 8]): 7 + 1 + 1 launches
        # per stage instead of 9 per source view -- at 512x640 a per-view launch is 10-20 us of a few hundred workgroups
        # (configuration 3: 162 -> 66 launches per forward); same kernels, same values
        # ... as long as a group's 8-channel volume stays well inside the 256 MiB Infinity Cache: between the layers of ONE view's
        # U-Net the activations are served from it, a batch of eight 118 MB volumes (configuration 5) is not -- measured there:
        # deconv 1.53 -> 1.98 ms when batched, the stage as a whole no faster -- so large volumes keep one launch per view
        n_s, n_b = costs.shape[0], costs.shape[1]
        h, w = costs.shape[3], costs.shape[4]
        group = max(1, min(n_s, self.PAIR_BATCH_BYTES // max(1, costs[0].numel() * costs.element_size())))
        # expected index and entropy of ALL pairs land in two buffers (each group's softargmin writes its batch slice), so the
        # per-pair `index * interval + start` (model_cas.py:348) is two launches per stage and the UncertNet reads its batch in place
        index_all = torch.empty((n_s * n_b, h, w), dtype=torch.float32, device=costs.device)
        entropy_all = torch.empty((n_s * n_b, h, w), dtype=torch.float32, device=costs.device)
        # PAIR_STREAMS > 1 (rounds 4 / 6 experiment; default 1): the per-view passes of a stage are independent until the fusion, so
        # consecutive groups may run on separate HIP streams (forked from / joined into the caller's stream; under the forward's
        # hipGraph capture they become parallel branches of the graph) -- the ramp of one view's kernels under the tail of another's
        n_streams = min(int(getattr(self, "PAIR_STREAMS", 1)), (n_s + group - 1) // group) if taps is None and costs.is_cuda else 1
        if n_streams > 1:
            pool = self.__dict__.setdefault("_pair_streams", {})
            side = pool.get(costs.device)
            if side is None or len(side) < n_streams:
                side = pool[costs.device] = [torch.cuda.Stream(device=costs.device) for _ in range(n_streams)]
            main_stream = torch.cuda.current_stream(costs.device)
            for st in side[:n_streams]:
                st.wait_stream(main_stream)
        for gi, g0 in enumerate(range(0, n_s, group)):
            g1 = min(n_s, g0 + group)
            ctx = torch.cuda.stream(side[gi % n_streams]) if n_streams > 1 else contextlib.nullcontext()
            with ctx:
                interm_all = self.reg(costs[g0:g1].view(((g1 - g0) * n_b,) + tuple(costs.shape[2:])))
                idx_g, ent_g = index_all[g0 * n_b:g1 * n_b], entropy_all[g0 * n_b:g1 * n_b]
                # head + expected index + entropy in ONE pass over the pair volume; the fp32 scores exist only for `taps`
                score_all = self.reg_pair.head_index_entropy(interm_all, idx_g, ent_g, want_scores=taps is not None) if self.fused_pair_head else None
                if score_all is None:
                    score_all = self.reg_pair(interm_all)                                  # fp32 [views * n, d, h, w]
                    ops.softargmin(score_all, None, want_index=True, want_entropy=True, into={"index": idx_g, "entropy": ent_g})
            for i in range(g0, g1):
                sl = slice((i - g0) * n_b, (i - g0 + 1) * n_b)
                interms.append(interm_all[sl])
                if taps is not None and i == 0:
                    taps.update(cost0=costs[0], interm0=interm_all[sl], score0=score_all[sl], entropy0=entropy_all[:n_b])
        if n_streams > 1:
            for st in side[:n_streams]:
                main_stream.wait_stream(st)
        est_all = index_all.view(n_s, n_b, 1, h, w) * depth_interval + depth_start       # [n_b,1,1,1] / [n_b,1,h,w] broadcast
        # the 2-D UncertNet (eval-mode BatchNorm: per-sample) runs ONCE on the entropy maps of all pairs stacked along the batch
        # axis: one fused launch per stage (csrc/uncert_net.hip); same values
        heads_all = self.uncert_net(entropy_all.unsqueeze(1))
        for i in range(len(srcs_feat)):
            heads = [hd[i * n_b:(i + 1) * n_b] for hd in heads_all]
            pair_results.append([est_all[i], heads])
            uncerts.append(heads[0].squeeze(1).to(torch.float32).contiguous())
            if taps is not None and i == 0:
                taps.update(uncert0=heads[0])
        if world == 1:
            fused = ops.fuse_pairs(interms, uncerts)                                   # model_cas.py:354-357,385-386
            score = self.reg_fuse(fused)
            o = ops.softargmin(score, None, want_index=True, want_conf=True, conf_mode=1, window=2.0)
            index, conf = o["index"], o["conf"]
        else:
            import torch.distributed as dist
            index, conf, fused, score = self._fuse_view_shard(interms, uncerts, self.view_group)
            # every rank reports the pair results of ALL views, in view order
            flat = torch.stack([torch.cat([ed, hd[0]], dim=1) for ed, hd in pair_results])   # [mine,n,2,h,w]
            slots = (n_views + world - 1) // world
            padded = torch.zeros((slots,) + tuple(flat.shape[1:]), dtype=flat.dtype, device=flat.device)
            padded[:flat.shape[0]] = flat
            gathered = [torch.empty_like(padded) for _ in range(world)]
            dist.all_gather(gathered, padded, group=self.view_group)
            pair_results = []
            for i in range(n_views):
                rec = gathered[i % world][i // world]
                pair_results.append([rec[:, 0:1], [rec[:, 1:2]]])
        est_depth = index.unsqueeze(1) * depth_interval + depth_start                  # model_cas.py:404-405
        if taps is not None:
            taps.update(fused=fused, score=score)
        return est_depth, conf.unsqueeze(1), pair_results


class Model(nn.Module):
    def __init__(self):
        super().__init__()
        self.feat_ext = FeatExt()
        self.stage1 = SingleStage()
        self.stage2 = SingleStage()
        self.stage3 = SingleStage()
