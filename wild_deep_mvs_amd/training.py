"""Training-mode execution of the hot path on the pscv engine.

``train.py`` / ``models/trainer.py`` of the reference call ``model(...)`` in ``train()`` mode and then
``loss.backward()`` (train.py:185-191, models/trainer.py:96-206).  The reference gets the backward from ATen
autograd; here the hot path is two ``torch.autograd.Function`` nodes whose forward AND backward are HIP launches:

* ``WarpCostFn``    features -> cost volume.  Backward = ``pscv_warp_cost_bwd`` (gradient to the feature maps only: the
                    sampling grid is built under ``no_grad`` in the reference, models/MVSNet/module.py:127).
* ``RegressFn``     cost volume -> 3-D U-Net with BATCH-statistics BatchNorm -> softmax over D -> depth.  Forward per
                    block: raw MFMA conv, ``pscv_bn_stats``, ``pscv_bn_act``; backward per block: ``pscv_bn_bwd_reduce``,
                    ``pscv_bn_bwd_apply``, ``pscv_conv3d_wgrad`` (MFMA over voxels) and the data gradient as the adjoint
                    convolution on the SAME forward kernels (Conv3d s1 <-> flipped ConvTranspose3d s1, Conv3d s2 <->
                    ConvTranspose3d s2 op1), with the gradient arriving over a skip connection added in that launch's epilogue.

PyTorch is plumbing: autograd graph, parameter storage, the handful of [C]-vector ops that turn sums into BatchNorm
coefficients, and the 2-D feature extractor upstream of the path (stays on PyTorch-ROCm in training).
Storage is 16-bit (bf16 by default in training: gradients span many decades), accumulation fp32.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn

from . import _lib as L
from . import ops


# --------------------------------------------------------------------------------------------------
# fused warp + cost with backward
# --------------------------------------------------------------------------------------------------
class WarpCostFn(torch.autograd.Function):
    """(ref NCHW fp32, src_0..src_{n-1} NCHW fp32 [, temp]) -> cost volume [B,D,h,w,C] in the storage dtype.

    The layout / precision conversion to the engine's channels-last 16-bit maps happens inside the node, so the
    gradients it returns are fp32 NCHW: the feature gradient is never rounded to 16 bits."""

    @staticmethod
    def forward(ctx, cams, depth, geom, cost, dtype, temp, ref, *srcs):
        # NCHW fp32 maps (the PyTorch-ROCm extractor) are converted here; [B,h,w,C] maps already in the storage dtype (the
        # engine's own extractor, FeatureNetFn) are taken as they are
        cl_in = ref.dtype == dtype and dtype != torch.float32
        if cl_in:
            ref_cl, srcs_cl = ref.detach().contiguous(), [s.detach().contiguous() for s in srcs]
        else:
            ref_cl = ops.to_channels_last(ref.detach(), dtype)
            srcs_cl = [ops.to_channels_last(s.detach(), dtype) for s in srcs]
        tval = float(temp.detach().float().item()) if temp is not None else 0.0
        out = ops.warp_cost(ref_cl, srcs_cl, cams, depth, geom=geom, cost=cost, temp=tval, out_dtype=dtype)
        ctx.save_for_backward(cams, depth, ref_cl, *srcs_cl)
        ctx.meta = (geom, cost, tval, temp is not None, ref.dtype, [s.dtype for s in srcs], cl_in)
        return out

    @staticmethod
    def backward(ctx, g):
        cams, depth, ref_cl, *srcs_cl = ctx.saved_tensors
        geom, cost, tval, has_temp, ref_dt, src_dts, cl_in = ctx.meta
        g = g.contiguous()
        dref, dsrcs, dtemp = ops.warp_cost_bwd(ref_cl, srcs_cl, cams, depth, g, geom=geom, cost=cost, temp=tval,
                                               want_dtemp=has_temp)
        if cl_in:
            gref, gsrcs = dref.to(ref_dt), [d.to(dt) for d, dt in zip(dsrcs, src_dts)]
        else:
            gref = dref.permute(0, 3, 1, 2).to(ref_dt)
            gsrcs = [d.permute(0, 3, 1, 2).to(dt) for d, dt in zip(dsrcs, src_dts)]
        return (None, None, None, None, None, dtemp if has_temp else None, gref, *gsrcs)


class WarpOnlyFn(torch.autograd.Function):
    """Function-level plane-sweep warp of ONE source map with autograd to the map (``homo_warping`` of
    models/MVSNet/module.py:111-169 and models/CVP_MVSNet/models/modules.py:74-128, ``homography_warping`` of
    models/VisMVSNet/homography.py:107-120): src NCHW fp32 -> [B,C,D,h,w] fp32; the grid carries no gradient."""

    @staticmethod
    def forward(ctx, cams, depth, geom, ref_hw, src):
        fea = ops.to_channels_last(src.detach(), torch.float32)
        vol = ops.warp_cost(None, [fea], cams, depth, geom=geom, cost=L.COST_WARP_ONLY, ref_hw=ref_hw, out_dtype=torch.float32)
        ctx.save_for_backward(cams, depth, fea)
        ctx.meta = (geom, tuple(ref_hw), src.dtype)
        return ops.to_channels_first(vol[0])

    @staticmethod
    def backward(ctx, g):
        cams, depth, fea = ctx.saved_tensors
        geom, ref_hw, dt = ctx.meta
        if fea.shape[3] not in (16, 32):
            raise NotImplementedError("pscv warp backward: 16 or 32 feature channels")
        g_cl = g.permute(0, 2, 3, 4, 1).to(torch.float32).contiguous().unsqueeze(0)        # [1,B,D,h,w,C]
        _, dsrcs, _ = ops.warp_cost_bwd(None, [fea], cams, depth, g_cl, geom=geom, cost=L.COST_WARP_ONLY, ref_hw=ref_hw)
        return None, None, None, None, dsrcs[0].permute(0, 3, 1, 2).to(dt)


class HomographyWarpFn(torch.autograd.Function):
    """``homography_warping(input, H)`` with arbitrary 3x3 matrices per batch item or per reference pixel and autograd to ``input``
    (models/VisMVSNet/homography.py:107-120: the sample positions are computed under no_grad, grid_sample differentiates its input,
    :101-102).  input NCHW -> [m,c,h,w] fp32; forward ``pscv_homography_warp``, backward ``pscv_homography_warp_bwd``."""

    @staticmethod
    def forward(ctx, H, ref_hw, src):
        Hc = H.detach().to(torch.float32).contiguous()
        out = ops.homography_warp(ops.to_channels_last(src.detach(), torch.float32), Hc, ref_hw)
        ctx.save_for_backward(Hc)
        ctx.meta = (tuple(src.shape[2:]), src.dtype)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        (Hc,) = ctx.saved_tensors
        src_hw, dt = ctx.meta
        g_cl = g.permute(0, 2, 3, 1).to(torch.float32).contiguous()                        # [m,h,w,c]
        return None, None, ops.homography_warp_bwd(g_cl, Hc, src_hw).permute(0, 3, 1, 2).to(dt)


# --------------------------------------------------------------------------------------------------
# 3-D U-Net in train() mode
# --------------------------------------------------------------------------------------------------
@dataclass
class Block:
    """One 3x3x3 block of a regulariser: conv (or transposed conv) [+ BatchNorm3d] [+ ReLU] [+ skip]."""
    name: str
    src: str                       # name of the input tensor ("cost" = the block input of the net)
    weight: torch.Tensor           # Conv3d [Co,Ci,3,3,3] / ConvTranspose3d [Ci,Co,3,3,3]
    stride: int = 1
    transposed: bool = False
    bn: Optional[nn.BatchNorm3d] = None
    relu: bool = True
    skip: Optional[str] = None     # tensor added AFTER the ReLU (models/MVSNet/model.py:79-81)
    conv_bias: Optional[torch.Tensor] = None   # only without bn (the `prob` head)

    @property
    def c_in(self):
        return int(self.weight.shape[0] if self.transposed else self.weight.shape[1])

    @property
    def c_out(self):
        return int(self.weight.shape[1] if self.transposed else self.weight.shape[0])

    @property
    def kind(self):
        if self.stride == 1:
            return L.CONV_S1
        return L.CONV_T2 if self.transposed else L.CONV_S2


def _fwd_layer(b: Block, dtype, dev) -> ops.Conv3dLayer:
    """Raw convolution (no folded statistics, no ReLU): the BatchNorm of a training step needs the un-normalised output."""
    return ops.Conv3dLayer.build(b.weight, kind=b.kind, transposed=b.transposed, device=dev, dtype=dtype,
                                 conv_bias=b.conv_bias if b.bn is None else None)


def _dgrad_layer(b: Block, dtype, dev) -> ops.Conv3dLayer:
    """Adjoint of the block's convolution as a forward layer of the engine, on the same weight tensor:
        Conv3d s1          [Co,Ci] -> ConvTranspose3d s1 weight with in = Co, out = Ci (flipped taps)
        Conv3d s2          [Co,Ci] -> ConvTranspose3d s2 p1 op1 (even input sizes: exact adjoint)
        ConvTranspose3d s1 [Ci,Co] -> Conv3d s1 with the taps flipped back = a ConvTranspose-packed conv on the
                                      channel-swapped weight
        ConvTranspose3d s2 [Ci,Co] -> Conv3d s2 weight with out = Ci, in = Co (same memory layout)."""
    w = b.weight
    if not b.transposed:
        kind = L.CONV_S1 if b.stride == 1 else L.CONV_T2
        return ops.Conv3dLayer.build(w, kind=kind, transposed=True, device=dev, dtype=dtype)
    if b.stride == 2:
        return ops.Conv3dLayer.build(w, kind=L.CONV_S2, transposed=False, device=dev, dtype=dtype)
    # stride-1 deconv: y[o] = sum_i x[i] w[ci,co,o-i+1]  ->  dx[i] = sum_o dy[o] w[ci,co,o-i+1]: a plain Conv3d with weight
    # [out = ci, in = co] and unflipped taps
    return ops.Conv3dLayer.build(w, kind=L.CONV_S1, transposed=False, device=dev, dtype=dtype)


def _bn_affine(b: Block, sums: torch.Tensor, nvox: int):
    """Batch statistics -> (scale, bias, mean, invstd); updates the running statistics like nn.BatchNorm3d.train().  One launch
    (pscv_bn_finalize): the dozen C-element tensor ops this used to be were a tenth of a Vis-MVSNet training step's wall time."""
    if not b.bn.training:
        return _bn_frozen_affine(b.bn)
    out = ops.bn_finalize(sums, nvox, b.bn)
    return out[0], out[1], out[2], out[3]


def _bn_frozen_affine(bn):
    """A BatchNorm submodule in eval() inside a training step (frozen-BN fine-tuning; nn.BatchNorm honours the flag per module):
    normalise with the running statistics, leave them and ``num_batches_tracked`` untouched.  The backward is then the eval-mode
    one, ``dy = gamma * invstd * dz`` (see ``_bn_coeffs``)."""
    if bn.running_mean is None or bn.running_var is None:
        raise NotImplementedError("pscv BatchNorm training: a BatchNorm in eval() without running statistics normalises with batch "
                                  "statistics in PyTorch; put it in train() instead")
    mean = bn.running_mean.detach().float()
    invstd = torch.rsqrt(bn.running_var.detach().float() + bn.eps)
    scale = invstd if bn.weight is None else bn.weight.detach().float() * invstd
    bias = -mean * scale if bn.bias is None else bn.bias.detach().float() - mean * scale
    return scale.contiguous(), bias.contiguous(), mean.contiguous(), invstd.contiguous()


def _bn_coeffs(bn, s, mean, invstd, nvox):
    """(ca, cb, cc, d gamma, d beta) of ``dy = ca dz + cb y + cc``: the batch-statistics backward, or for a frozen BatchNorm
    (eval() inside a training step) the eval-mode one, where the statistics are constants: cb = cc = 0."""
    co = ops.bn_bwd_coeffs(s, mean, invstd, bn.weight, nvox)
    if not bn.training:
        co[1:3].zero_()
    return co


# Test hook: when set to a dict, RegressFn records every block's forward and backward tensors in it (the parity tests
# replay each block through ATen autograd on exactly these tensors: random-weight BatchNorm nets amplify any 16-bit
# rounding difference by ~3x per layer, so only per-block comparisons on shared inputs are sharp).
TRACE: Optional[dict] = None


class RegressFn(torch.autograd.Function):
    """cost volume -> train()-mode 3-D U-Net -> softmax over D -> (depth, photometric confidence).

    ``forward(ctx, blocks, out_name, depth_values, dtype, cost, *params)``: ``blocks`` is the topologically ordered
    block list of the regulariser (the last one is the 1-channel ``prob`` head), ``params`` the tensors autograd
    must see, in the order ``block_params(blocks)`` lists them."""

    @staticmethod
    def block_params(blocks: Sequence[Block]) -> List[torch.Tensor]:
        ps = []
        for b in blocks:
            ps.append(b.weight)
            if b.bn is not None:
                ps += [b.bn.weight, b.bn.bias]
            elif b.conv_bias is not None:
                ps.append(b.conv_bias)
        return ps

    @staticmethod
    def forward(ctx, blocks, depth_values, dtype, cost, *params):
        dev = cost.device
        t: Dict[str, torch.Tensor] = {"cost": cost}
        saved: Dict[str, tuple] = {}
        for b in blocks[:-1]:
            x = t[b.src]
            y = ops.conv3d(x, _fwd_layer(b, dtype, dev))
            nvox = y.numel() // y.shape[4]
            scale, bias, mean, invstd = _bn_affine(b, ops.bn_stats(y), nvox)
            t[b.name] = ops.bn_act(y, scale, bias, relu=b.relu, skip=t[b.skip] if b.skip else None)
            saved[b.name] = (y, scale, bias, mean, invstd)
            if TRACE is not None:
                TRACE[b.name] = dict(x=x, y=y, act=t[b.name], skip=t[b.skip] if b.skip else None, mean=mean, invstd=invstd)
        head = blocks[-1]
        logits = ops.conv3d(t[head.src], _fwd_layer(head, dtype, dev), out_dtype=torch.float32)
        B, D, h, w, _ = logits.shape
        logits = logits.view(B, D, h, w)
        o = ops.softargmin(logits, depth_values, want_conf=True, conf_mode=0)
        ctx.blocks, ctx.t, ctx.saved_bn, ctx.dtype = blocks, t, saved, dtype
        ctx.logits, ctx.depth_values = logits, depth_values
        ctx.mark_non_differentiable(o["conf"])
        if TRACE is not None:
            TRACE[head.name] = dict(x=t[head.src], logits=logits)
        return o["depth"], o["conf"]

    @staticmethod
    def backward(ctx, g_depth, _g_conf):
        blocks, t, saved, dtype = ctx.blocks, ctx.t, ctx.saved_bn, ctx.dtype
        dev = ctx.logits.device
        grads: Dict[str, Optional[torch.Tensor]] = {}
        pgrads: Dict[int, torch.Tensor] = {}

        def push(name: str, g: torch.Tensor):
            grads[name] = g if grads.get(name) is None else grads[name] + g

        def dgrad(b: Block, dy: torch.Tensor):
            """Gradient of block b's convolution w.r.t. its input, added to whatever already arrived for that tensor."""
            lay = _dgrad_layer(b, dtype, dev)
            prev = grads.get(b.src)
            grads[b.src] = ops.conv3d(dy, lay, skip=prev)

        def wgrad(b: Block, x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
            if b.transposed:   # [Ci,Co,27]: P = layer input, Q = grad of the output
                return ops.conv3d_wgrad(x, dy, ca=b.c_in, cb=dy.shape[4], stride=b.stride)
            return ops.conv3d_wgrad(dy, x, ca=dy.shape[4], cb=b.c_in, stride=b.stride)

        # softmax + regression: d loss / d logits, in the 8-channel layout of the conv kernels (channel 0 carries it)
        head = blocks[-1]
        dl8 = ops.softargmin_bwd(ctx.logits, ctx.depth_values, g_depth.contiguous().float(), dtype)
        x_head = t[head.src]
        dw = wgrad(head, x_head, dl8)                              # [8, Ci, 27], row 0 is the head's filter
        pgrads[id(head.weight)] = dw[:1].to(head.weight.dtype)
        if head.conv_bias is not None:
            pgrads[id(head.conv_bias)] = ops.bn_stats(dl8)[0, :1].to(head.conv_bias.dtype)
        # data gradient of the head: ConvTranspose-packed conv from the zero-padded 8 "output channels" back to Ci
        w8 = torch.zeros((8,) + tuple(head.weight.shape[1:]), dtype=torch.float32, device=dev)
        w8[:1] = head.weight.detach().float()
        lay = ops.Conv3dLayer.build(w8, kind=L.CONV_S1, transposed=True, device=dev, dtype=dtype)
        grads[head.src] = ops.conv3d(dl8, lay)
        if TRACE is not None:
            TRACE[head.name].update(dl8=dl8, dw=pgrads[id(head.weight)], dbias=pgrads.get(id(head.conv_bias)), dx=grads[head.src])
        del dl8

        for b in reversed(blocks[:-1]):
            dact = grads.pop(b.name)
            if b.skip:
                push(b.skip, dact)
            y, scale, bias, mean, invstd = saved[b.name]
            nvox = y.numel() // y.shape[4]
            s = ops.bn_bwd_reduce(dact, y, scale, bias, relu=b.relu)
            ca, cb, cc, s2, s1 = _bn_coeffs(b.bn, s, mean, invstd, nvox)     # s2 = sum dz * xhat, s1 = sum dz
            dy = ops.bn_bwd_apply(dact, y, scale, bias, ca, cb, cc, relu=b.relu)
            pgrads[id(b.bn.weight)] = s2.to(b.bn.weight.dtype)
            pgrads[id(b.bn.bias)] = s1.to(b.bn.bias.dtype)
            pgrads[id(b.weight)] = wgrad(b, t[b.src], dy).to(b.weight.dtype)
            prev = grads.get(b.src)
            if b.src != "cost" or ctx.needs_input_grad[3]:
                dgrad(b, dy)
            if TRACE is not None:
                TRACE[b.name].update(dact=dact, dy=dy, dgamma=s2, dbeta=s1, dw=pgrads[id(b.weight)], dx_prev=prev, dx=grads.get(b.src))
            del dact, dy

        out = [None, None, None, grads.get("cost")]
        for p in RegressFn.block_params(blocks):
            out.append(pgrads.get(id(p)))
        ctx.t = ctx.saved_bn = None
        return tuple(out)


# --------------------------------------------------------------------------------------------------
# Vis-MVSNet: residual-block U-Net, score heads, visibility-weighted fusion
# --------------------------------------------------------------------------------------------------
def _bn_stats_affine(bn: nn.BatchNorm3d, y: torch.Tensor, groups: int = 1):
    """Batch statistics of a stored conv output -> (scale, bias, mean, invstd, nvox); updates the running statistics.  ``groups`` > 1:
    the batch axis holds that many consecutive slices (the source views of a sample), each normalised with its own statistics like
    the reference's per-view calls; the constants come back as [G,C] rows and ``nvox`` counts one group."""
    if groups == 1:
        nvox = y.numel() // y.shape[4]
        scale, bias, mean, invstd = _bn_affine(Block("", "", y, bn=bn), ops.bn_stats(y), nvox)
        return scale, bias, mean, invstd, nvox
    nvox = y.numel() // y.shape[4] // groups
    aff = _bn_affine_grouped(bn, ops.bn_stats(y, groups), nvox, groups)
    return aff[:, 0], aff[:, 1], aff[:, 2], aff[:, 3], nvox


def _bn_backward(bn: nn.BatchNorm3d, dz_src: torch.Tensor, y: torch.Tensor, saved, relu: bool):
    """BatchNorm(+ReLU before any skip) backward on the engine: returns (dy, d gamma, d beta); grouped constants ([G,C] rows of
    ``_bn_stats_affine(..., groups)``) give per-group coefficients and gradients summed over the groups (one shared module)."""
    scale, bias, mean, invstd, nvox = saved
    s = ops.bn_bwd_reduce(dz_src, y, scale, bias, relu=relu)
    if scale.dim() == 2:
        co = ops.bn_bwd_coeffs(s, mean, invstd, bn.weight, nvox)                       # [G,5,C]
        if not bn.training:
            co[:, 1:3].zero_()
        dy = ops.bn_bwd_apply(dz_src, y, scale, bias, co[:, 0], co[:, 1], co[:, 2], relu=relu)
        dgb = co[:, 3:5].sum(0)
        return dy, dgb[0].to(bn.weight.dtype), dgb[1].to(bn.bias.dtype)
    ca, cb, cc, s2, s1 = _bn_coeffs(bn, s, mean, invstd, nvox)
    dy = ops.bn_bwd_apply(dz_src, y, scale, bias, ca, cb, cc, relu=relu)
    return dy, s2.to(bn.weight.dtype), s1.to(bn.bias.dtype)


class VisUNetFn(torch.autograd.Function):
    """``UNet(8, 1, 0, 4, [], [8, 16], [], tag, dim=3)`` of Vis-MVSNet (models/VisMVSNet/nn_utils.py:194-278, used by
    ``Reg`` / ``RegFuse``, model_cas.py:38-74) in train() mode, forward and backward on the engine:

        enc0 = relu(bn(conv(relu(bn(conv(x))))) + x)                         BasicBlock 8 -> 8
        e1   = relu(bn(conv(relu(bn(conv_s2(enc0))))) + bn(conv1x1_s2(enc0)))  BasicBlock 8 -> 16, strided 1x1x1 shortcut
        out  = conv(cat([deconv_s2(e1), enc0]))                              linear decoder (no BN / ReLU)

    ``forward(ctx, holder, dtype, x, *params)`` with ``params = VisUNetFn.params(holder)``; x / out [n,d,h,w,8] 16-bit."""

    @staticmethod
    def parts(holder):
        enc = list(holder.unet.enc_blocks.values())
        return enc[0][0], enc[1][0], list(holder.unet.dec_blocks.values())[0]

    @staticmethod
    def params(holder) -> List[torch.Tensor]:
        b0, b1, dec = VisUNetFn.parts(holder)
        return [b0.conv1.weight, b0.bn1.weight, b0.bn1.bias, b0.conv2.weight, b0.bn2.weight, b0.bn2.bias,
                b1.conv1.weight, b1.bn1.weight, b1.bn1.bias, b1.downsample[0].weight, b1.downsample[1].weight, b1.downsample[1].bias,
                b1.conv2.weight, b1.bn2.weight, b1.bn2.bias, dec[0].weight, dec[1].weight]

    @staticmethod
    def _ds_weight(b1):
        """The strided 1x1x1 shortcut conv as the centre tap of a 3x3x3 stride-2 conv (reads 2*o + 1 - 1 = 2*o)."""
        w = b1.downsample[0].weight
        w3 = torch.zeros((w.shape[0], w.shape[1], 3, 3, 3), dtype=torch.float32, device=w.device)
        w3[:, :, 1, 1, 1] = w.detach().float().view(w.shape[0], w.shape[1])
        return w3

    @staticmethod
    def forward(ctx, holder, dtype, groups, x, *params):
        # groups > 1: the batch axis is that many consecutive slices -- the source views of a sample, whose pair U-Nets the reference
        # runs one view at a time (model_cas.py:341-352) -- each with its own BatchNorm statistics, in ONE launch per layer
        b0, b1, dec = VisUNetFn.parts(holder)
        dev = x.device
        _bsa = lambda bn, y: _bn_stats_affine(bn, y, groups)
        mk = lambda w, kind, tr=False: ops.Conv3dLayer.build(w, kind=kind, transposed=tr, device=dev, dtype=dtype)
        y1 = ops.conv3d(x, mk(b0.conv1.weight, L.CONV_S1))
        a1 = _bsa(b0.bn1, y1)
        t = ops.bn_act(y1, a1[0], a1[1], relu=True)
        y2 = ops.conv3d(t, mk(b0.conv2.weight, L.CONV_S1))
        a2 = _bsa(b0.bn2, y2)
        enc0 = ops.bn_act(y2, a2[0], a2[1], relu="post", skip=x)
        y3 = ops.conv3d(enc0, mk(b1.conv1.weight, L.CONV_S2))
        a3 = _bsa(b1.bn1, y3)
        t1 = ops.bn_act(y3, a3[0], a3[1], relu=True)
        y4 = ops.conv3d(enc0, mk(VisUNetFn._ds_weight(b1), L.CONV_S2))
        a4 = _bsa(b1.downsample[1], y4)
        ds = ops.bn_act(y4, a4[0], a4[1], relu=False)
        y5 = ops.conv3d(t1, mk(b1.conv2.weight, L.CONV_S1))
        a5 = _bsa(b1.bn2, y5)
        e1 = ops.bn_act(y5, a5[0], a5[1], relu="post", skip=ds)
        up = ops.conv3d(e1, mk(dec[0].weight, L.CONV_T2, True))
        cat = torch.cat([up, enc0], dim=4)                                   # deconv channels first (nn_utils.py:269-271)
        out = ops.conv3d(cat, mk(dec[1].weight, L.CONV_S1))
        ctx.holder, ctx.dtype = holder, dtype
        ctx.t = dict(x=x, y1=y1, t=t, y2=y2, enc0=enc0, y3=y3, t1=t1, y4=y4, y5=y5, e1=e1, cat=cat)
        ctx.a = (a1, a2, a3, a4, a5)
        if TRACE is not None:
            TRACE.setdefault("vis_unet", []).append(dict(ctx.t, ds=ds, up=up, out=out))
        return out

    @staticmethod
    def backward(ctx, g):
        b0, b1, dec = VisUNetFn.parts(ctx.holder)
        T_, dtype = ctx.t, ctx.dtype
        a1, a2, a3, a4, a5 = ctx.a
        dev = g.device
        g = g.contiguous()
        mk = lambda w, kind, tr=False: ops.Conv3dLayer.build(w, kind=kind, transposed=tr, device=dev, dtype=dtype)
        # post conv 16 -> 8 (linear): weight gradient, then d cat (16 channels) through the flipped-tap adjoint
        dw_post = ops.conv3d_wgrad(g, T_["cat"], ca=8, cb=16, stride=1)
        dcat = ops.conv3d(g, mk(dec[1].weight, L.CONV_S1, True))
        # deconv 16 -> 8 s2 (linear): P = its input e1, Q = channels 0..7 of d cat
        dw_dec = ops.conv3d_wgrad(T_["e1"], dcat, ca=16, cb=8, stride=2)
        de1 = ops.conv3d(dcat, mk(dec[0].weight, L.CONV_S2, False), in_coff=0)
        # block 1, second conv: e1 = relu(bn2(conv2(t1)) + ds)
        dpre = ops.relu_bwd(de1, T_["e1"])
        dy5, dg5, db5 = _bn_backward(b1.bn2, dpre, T_["y5"], a5, relu=False)
        dw_c2b = ops.conv3d_wgrad(dy5, T_["t1"], ca=16, cb=16, stride=1)
        dt1 = ops.conv3d(dy5, mk(b1.conv2.weight, L.CONV_S1, True))
        # shortcut: ds = bn(conv1x1_s2(enc0)); its gradient is dpre
        dy4, dg4, db4 = _bn_backward(b1.downsample[1], dpre, T_["y4"], a4, relu=False)
        dw_ds = ops.conv3d_wgrad(dy4, T_["enc0"], ca=16, cb=8, stride=2)[:, :, 1, 1, 1].reshape(b1.downsample[0].weight.shape)
        denc0 = ops.conv3d(dy4, mk(VisUNetFn._ds_weight(b1), L.CONV_T2, True), skip=dcat, skip_coff=8)   # + d cat[..., 8:16]
        # block 1, first conv (stride 2)
        dy3, dg3, db3 = _bn_backward(b1.bn1, dt1, T_["y3"], a3, relu=True)
        dw_c1b = ops.conv3d_wgrad(dy3, T_["enc0"], ca=16, cb=8, stride=2)
        denc0 = ops.conv3d(dy3, mk(b1.conv1.weight, L.CONV_T2, True), skip=denc0)
        # block 0: enc0 = relu(bn2(conv2(t)) + x)
        dpre0 = ops.relu_bwd(denc0, T_["enc0"])
        dy2, dg2, db2 = _bn_backward(b0.bn2, dpre0, T_["y2"], a2, relu=False)
        dw_c2a = ops.conv3d_wgrad(dy2, T_["t"], ca=8, cb=8, stride=1)
        dt = ops.conv3d(dy2, mk(b0.conv2.weight, L.CONV_S1, True))
        dy1, dg1, db1 = _bn_backward(b0.bn1, dt, T_["y1"], a1, relu=True)
        dw_c1a = ops.conv3d_wgrad(dy1, T_["x"], ca=8, cb=8, stride=1)
        dx = ops.conv3d(dy1, mk(b0.conv1.weight, L.CONV_S1, True), skip=dpre0) if ctx.needs_input_grad[3] else None
        if TRACE is not None:
            TRACE.setdefault("vis_unet_bwd", []).append(dict(g=g, dcat=dcat, de1=de1, dpre=dpre, dy5=dy5, dt1=dt1, dy4=dy4, dy3=dy3,
                                                             denc0=denc0, dpre0=dpre0, dy2=dy2, dt=dt, dy1=dy1, dx=dx))
        wd = lambda t_, p: t_.to(p.dtype)
        grads = [wd(dw_c1a, b0.conv1.weight), dg1, db1, wd(dw_c2a, b0.conv2.weight), dg2, db2,
                 wd(dw_c1b, b1.conv1.weight), dg3, db3, wd(dw_ds, b1.downsample[0].weight), dg4, db4,
                 wd(dw_c2b, b1.conv2.weight), dg5, db5, wd(dw_dec, dec[0].weight), wd(dw_post, dec[1].weight)]
        ctx.t = None
        return (None, None, None, dx, *grads)


class ScoreHeadFn(torch.autograd.Function):
    """1-channel score head (``final_conv`` 8 -> 1, no bias: RegPair / RegFuse, model_cas.py:51-74) + softmax over the
    planes + expected index, entropy and the +-window probability (nn_utils.py:453-470), forward and backward on the
    engine.  Returns (index [n,h,w], entropy [n,h,w], window prob [n,h,w] or None-like zeros); the window probability is
    not differentiated (the reference only reports it)."""

    @staticmethod
    def forward(ctx, dtype, window, interm, weight):
        lay = ops.Conv3dLayer.build(weight, kind=L.CONV_S1, device=interm.device, dtype=dtype)
        score = ops.conv3d(interm, lay, out_dtype=torch.float32).squeeze(-1)
        o = ops.softargmin(score, None, want_index=True, want_entropy=True, want_conf=window is not None, conf_mode=1,
                           window=float(window or 0))
        ctx.dtype, ctx.score, ctx.interm, ctx.weight = dtype, score, interm, weight
        conf = o["conf"] if window is not None else torch.zeros_like(o["index"])
        ctx.mark_non_differentiable(conf)
        if TRACE is not None:
            TRACE.setdefault("vis_head", []).append(dict(interm=interm, score=score, index=o["index"], entropy=o["entropy"]))
        return o["index"], o["entropy"], conf

    @staticmethod
    def backward(ctx, g_index, g_entropy, _g_conf):
        dtype, score, interm, weight = ctx.dtype, ctx.score, ctx.interm, ctx.weight
        gi = g_index.contiguous().float() if g_index is not None else None
        ge = g_entropy.contiguous().float() if g_entropy is not None else None
        dl8 = ops.softargmin_bwd(score, None, None, dtype, grad_index=gi, grad_entropy=ge)
        dw = ops.conv3d_wgrad(dl8, interm, ca=8, cb=8, stride=1)[:1].to(weight.dtype)
        w8 = torch.zeros((8,) + tuple(weight.shape[1:]), dtype=torch.float32, device=interm.device)
        w8[:1] = weight.detach().float()
        dinterm = ops.conv3d(dl8, ops.Conv3dLayer.build(w8, kind=L.CONV_S1, transposed=True, device=interm.device, dtype=dtype))
        if TRACE is not None:
            TRACE.setdefault("vis_head_bwd", []).append(dict(dl8=dl8, dw=dw, dinterm=dinterm))
        return None, None, dinterm, dw


class FusePairsFn(torch.autograd.Function):
    """fused = sum_v exp(-u_v) I_v / sum_v exp(-u_v) (model_cas.py:354-357,385-386), forward and backward on the engine.
    ``forward(ctx, n, I_0..I_{n-1}, u_0..u_{n-1})`` with I_v [B,D,h,w,8] 16-bit and u_v [B,h,w] fp32."""

    @staticmethod
    def forward(ctx, n, *tensors):
        interms = [t.contiguous() for t in tensors[:n]]
        uncerts = [t.contiguous().float() for t in tensors[n:]]
        ctx.n, ctx.interms, ctx.uncerts = n, interms, uncerts
        return ops.fuse_pairs(interms, uncerts)

    @staticmethod
    def backward(ctx, g):
        dI, dU = ops.fuse_pairs_bwd(ctx.interms, ctx.uncerts, g.contiguous())
        return (None, *dI, *dU)


# --------------------------------------------------------------------------------------------------
# MVSNet's 2-D FeatureNet in train() mode on the engine (upstream of the path; optional)
# --------------------------------------------------------------------------------------------------
def _v5(x: torch.Tensor) -> torch.Tensor:
    """[B,H,W,C] -> [B,1,H,W,C]: the 3-D kernels (batch statistics, weight gradient with a single plane) on 2-D maps."""
    return x.unsqueeze(1)


def _wgrad2d_k3(dy: torch.Tensor, x: torch.Tensor, co: int, ci: int) -> torch.Tensor:
    """Weight gradient of a k3 s1 p1 Conv2d as the single-plane case of the 3-D MFMA kernel (its z taps 0 and 2 only meet
    zero padding): [co,ci,3,3]."""
    return ops.conv3d_wgrad(_v5(dy), _v5(x), ca=co, cb=ci, stride=1)[:, :, 1]


_K5_IDX = {}


def _wgrad2d_k5s2(dy: torch.Tensor, x: torch.Tensor, co: int, ci: int) -> torch.Tensor:
    """Weight gradient of a k5 s2 p2 Conv2d: x[2o + k - 2] with k = 2 t + a is tap t - 1 of the input's parity plane
    x_a[i] = x[2i + a], so the 5x5 gradient is assembled from four stride-1 3x3 weight gradients -- with one indexed gather
    (kernel index k -> parity k % 2, tap k // 2)."""
    G = torch.stack([_wgrad2d_k3(dy, x[:, a::2, b::2, :].contiguous(), co, ci) for a in (0, 1) for b in (0, 1)])       # [4,co,ci,3,3]
    idx = _K5_IDX.get(dy.device)
    if idx is None:
        P = [[(ky % 2) * 2 + (kx % 2) for kx in range(5)] for ky in range(5)]
        TY = [[ky // 2 for kx in range(5)] for ky in range(5)]
        TX = [[kx // 2 for kx in range(5)] for ky in range(5)]
        idx = _K5_IDX[dy.device] = tuple(torch.tensor(t_, dtype=torch.long, device=dy.device) for t_ in (P, TY, TX))
    P, TY, TX = idx
    return G[P, :, :, TY, TX].permute(2, 3, 0, 1).contiguous()                       # [5,5,co,ci] -> [co,ci,5,5]


def _dgrad2d_k3_layer(w: torch.Tensor, dtype) -> ops.Conv2dLayer:
    """Adjoint of a k3 s1 p1 Conv2d [Co,Ci,3,3]: the same conv kernel on the channel-swapped, tap-flipped weight."""
    return ops.Conv2dLayer.build(w, stride=1, dtype=dtype, adjoint=True)        # (packed in place from the forward weight: no flip / transpose launches)


def _dgrad2d_k5s2_layers(w: torch.Tensor, dtype) -> List[ops.Conv2dLayer]:
    """Adjoint of a k5 s2 p2 Conv2d [Co,Ci,5,5] (= ConvTranspose2d k5 s2 p2 op1) as four stride-1 3x3 sub-convolutions, one per
    output parity: dx[2i + a] = sum_m dy[i + m] w[k = a + 2 - 2m], m = -1, 0, +1 (k = 5 does not exist: zero)."""
    wf = w.detach().float()
    # kernel index per (parity a, tap t): k = a + 4 - 2 t, none (a zero plane appended at index 5) outside 0..4; one indexed gather
    wp = torch.nn.functional.pad(wf, (0, 1, 0, 1))                                  # [co,ci,6,6]
    k = _K5_IDX.get(("d", wf.device))
    if k is None:
        k = _K5_IDX[("d", wf.device)] = torch.tensor([[a + 4 - 2 * t_ if 0 <= a + 4 - 2 * t_ <= 4 else 5 for t_ in range(3)] for a in (0, 1)],
                                                     dtype=torch.long, device=wf.device)
    W = wp[:, :, k[:, :, None, None], k[None, None, :, :]]                          # [co,ci,a,ty,b,tx]
    return [ops.Conv2dLayer.build(W[:, :, a, :, b, :].transpose(0, 1).contiguous(), stride=1, dtype=dtype) for a in (0, 1) for b in (0, 1)]


def _cached_layer(net, tag, weight, dtype, make, extra=()):
    """Packed 2-D layers are built on the host (pscv_pack_conv2d_weights): keep them per (weight version, dtype) on the module so
    that the views of one step -- same weights -- do not repack (and synchronise) once per view."""
    cache = net.__dict__.setdefault("_pscv_train_layers", {})
    key = (tag, ops.weights_epoch(), weight.data_ptr(), weight._version, dtype) + tuple(extra)     # extra: e.g. a folded bias' version
    if key not in cache:
        # one live entry per tag: an optimiser step bumps weight._version, the previous step's packed layer goes here
        for k in [k for k in cache if k[0] == tag]:
            del cache[k]
        cache[key] = make()
    return cache[key]


def _bn_affine_grouped(bn, sums: torch.Tensor, nvox: int, groups: int) -> torch.Tensor:
    """[G,4,C] = (scale, bias, mean, invstd) per group from grouped batch statistics [G,2,C]; a frozen BatchNorm (eval() inside a
    training step) hands every group the running statistics."""
    if not bn.training:
        return torch.stack(_bn_frozen_affine(bn), 0).unsqueeze(0).expand(groups, -1, -1).contiguous()
    return ops.bn_finalize(sums.view(groups, 2, -1), nvox, bn)


class FeatureNetFn(torch.autograd.Function):
    """MVSNet's 2-D extractor (models/MVSNet/model.py:21-41: seven conv + BatchNorm2d + ReLU blocks, k3 s1 / k5 s2, and a
    final k3 conv with bias) in train() mode, forward and backward on the engine: raw MFMA conv2d, batch statistics, one
    normalise + ReLU pass; backward = BatchNorm backward, weight gradient on the MFMA weight-gradient kernel (single-plane
    mode; k5 s2 from the four parity planes of the input), data gradient as the adjoint conv on the forward kernel (k5 s2: four
    parity sub-convolutions).  ``forward(ctx, net, dtype, groups, img [G*B,3,H,W], *params)`` -> [G*B,H/4,W/4,32] in ``dtype``.

    ``groups``: the images are ``groups`` consecutive slices of B images, each normalised with its OWN batch statistics -- the
    views of a sample, which the reference sends through the extractor one at a time (model.py:101-107), here in ONE launch per
    pass (the convolutions and weight gradients do not care; the BatchNorm passes take a group axis, pscv_bn_*_grouped) and the
    running statistics see the views in order.  groups = 1 is one plain call of the module."""

    @staticmethod
    def params(net) -> List[torch.Tensor]:
        ps = []
        for i in range(7):
            blk = getattr(net, f"conv{i}")
            ps += [blk.conv.weight, blk.bn.weight, blk.bn.bias]
        return ps + [net.feature.weight, net.feature.bias]

    @staticmethod
    def forward(ctx, net, dtype, groups, img, *params):
        GB, _, H, W = img.shape
        if H % 4 or W % 4:
            raise ValueError("pscv FeatureNetFn: image height and width must be multiples of 4")
        if groups < 1 or GB % groups:
            raise ValueError(f"pscv FeatureNetFn: {GB} images do not split into {groups} groups")
        x = ops.image_to_channels_last8(img.detach(), dtype)
        saved = []
        for i, (ci, co, k, s_, p_) in enumerate(net.SPEC):
            blk = getattr(net, f"conv{i}")
            y = ops.conv2d(x, _cached_layer(net, f"f{i}", blk.conv.weight, dtype,
                                            lambda: ops.Conv2dLayer.build(blk.conv.weight, stride=s_, dtype=dtype)))
            nvox = y.numel() // y.shape[3] // groups
            aff = _bn_affine_grouped(blk.bn, ops.bn_stats(_v5(y), groups), nvox, groups)
            act = ops.bn_act(_v5(y), aff[:, 0], aff[:, 1], relu=True).squeeze(1)
            saved.append((x, y, aff, nvox))
            x = act
        out = ops.conv2d(x, ops.Conv2dLayer.build(net.feature.weight, stride=1, conv_bias=net.feature.bias, dtype=dtype))   # (bias may change alone)
        ctx.net, ctx.dtype, ctx.saved, ctx.x_last = net, dtype, saved, x
        return out

    @staticmethod
    def backward(ctx, g):
        net, dtype, saved = ctx.net, ctx.dtype, ctx.saved
        g = g.contiguous().to(dtype)
        grads = {}
        # final conv (bias, no BatchNorm)
        grads[id(net.feature.weight)] = _wgrad2d_k3(g, ctx.x_last, 32, 32).to(net.feature.weight.dtype)
        grads[id(net.feature.bias)] = ops.bn_stats(_v5(g))[0].to(net.feature.bias.dtype)
        dact = ops.conv2d(g, _cached_layer(net, "dfeat", net.feature.weight, dtype, lambda: _dgrad2d_k3_layer(net.feature.weight, dtype)))
        for i in range(6, -1, -1):
            ci, co, k, s_, p_ = net.SPEC[i]
            blk = getattr(net, f"conv{i}")
            x, y, aff, nvox = saved[i]
            s = ops.bn_bwd_reduce(_v5(dact), _v5(y), aff[:, 0], aff[:, 1], relu=True)
            cf = ops.bn_bwd_coeffs(s, aff[:, 2], aff[:, 3], blk.bn.weight, nvox)          # [G,5,C]
            if not blk.bn.training:
                cf[:, 1:3].zero_()
            dy = ops.bn_bwd_apply(_v5(dact), _v5(y), aff[:, 0], aff[:, 1], cf[:, 0], cf[:, 1], cf[:, 2], relu=True).squeeze(1)
            dgb = cf[:, 3:5].sum(0)                                                        # the views share the BatchNorm: gradients add
            cpad = x.shape[3]
            dw = _wgrad2d_k3(dy, x, co, cpad) if k == 3 else _wgrad2d_k5s2(dy, x, co, cpad)
            grads[id(blk.conv.weight)] = dw[:, :ci].to(blk.conv.weight.dtype)
            grads[id(blk.bn.weight)], grads[id(blk.bn.bias)] = dgb[0].to(blk.bn.weight.dtype), dgb[1].to(blk.bn.bias.dtype)
            if i > 0:
                if k == 3:
                    dact = ops.conv2d(dy, _cached_layer(net, f"d{i}", blk.conv.weight, dtype,
                                                        lambda: _dgrad2d_k3_layer(blk.conv.weight, dtype)))
                else:
                    dact = torch.empty_like(x)
                    subs = _cached_layer(net, f"d{i}", blk.conv.weight, dtype, lambda: _dgrad2d_k5s2_layers(blk.conv.weight, dtype))
                    for par, sub in enumerate(subs):
                        ops.conv2d(dy, sub, out=dact, parity=par)
        ctx.saved = None
        return (None, None, None, None, *[grads[id(p)] for p in FeatureNetFn.params(net)])


class FeaturePyramidFn(torch.autograd.Function):
    """CVP-MVSNet's 2-D pyramid tower (models/CVP_MVSNet/models/net.py:21-47: nine conv k3 + bias + LeakyReLU(0.1) layers, 3 -> 64
    -> 64 -> 64 -> 32 -> 32 -> 32 -> 16 -> 16 -> 16, applied to the image and to its bilinear half-scale copies) in train()
    mode, forward and backward on the engine, for ALL views at once (no BatchNorm: the views are plain batch items):
    forward = the eval-mode launches (MFMA conv2d with the bias and the LeakyReLU fused) keeping every layer's output; backward per
    level and layer = LeakyReLU backward from the stored OUTPUT (same sign as the input), bias gradient = per-channel sum,
    weight gradient on the single-plane mode of the MFMA weight-gradient kernel, data gradient = the adjoint conv on the forward
    kernel.  ``forward(ctx, tower, dtype, nscale, img [N,3,H,W], *params)`` -> nscale maps [N,H_l,W_l,16] in ``dtype``, finest
    first; the image pyramid carries no gradient (net.py:44 detaches it)."""

    SLOPE = 0.1

    @staticmethod
    def params(tower) -> List[torch.Tensor]:
        ps = []
        for n in tower._names:
            ps += [getattr(tower, n)[0].weight, getattr(tower, n)[0].bias]
        return ps

    @staticmethod
    def forward(ctx, tower, dtype, nscale, img, *params):
        convs = [getattr(tower, n)[0] for n in tower._names]
        layers = [_cached_layer(tower, f"p{i}", c.weight, dtype,
                                lambda c=c: ops.Conv2dLayer.build(c.weight, stride=1, conv_bias=c.bias, leaky=FeaturePyramidFn.SLOPE, dtype=dtype),
                                extra=(c.bias.data_ptr(), c.bias._version))          # (the bias is folded into the packed layer)
                  for i, c in enumerate(convs)]
        saved, outs = [], []
        for x in ops.image_pyramid_cl8(img.detach(), nscale, dtype):
            acts = [x]
            for layer in layers:
                acts.append(ops.conv2d(acts[-1], layer))
            saved.append(acts)
            outs.append(acts[-1])
        ctx.tower, ctx.dtype, ctx.saved = tower, dtype, saved
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        tower, dtype, saved = ctx.tower, ctx.dtype, ctx.saved
        convs = [getattr(tower, n)[0] for n in tower._names]
        dW = [None] * len(convs)
        dB = [None] * len(convs)
        for acts, g in zip(saved, gs):
            if g is None:
                continue
            g = g.contiguous().to(dtype)
            for i in range(len(convs) - 1, -1, -1):
                c = convs[i]
                co, ci = int(c.weight.shape[0]), int(c.weight.shape[1])
                x, out = acts[i], acts[i + 1]
                dpre, db = ops.relu_bwd_sum(_v5(g), _v5(out), FeaturePyramidFn.SLOPE)       # LeakyReLU backward + bias gradient in one pass
                dpre = dpre.squeeze(1)
                dw = _wgrad2d_k3(dpre, x, co, x.shape[3])[:, :ci]
                dW[i] = dw if dW[i] is None else dW[i] + dw
                dB[i] = db if dB[i] is None else dB[i] + db
                if i > 0:
                    g = ops.conv2d(dpre, _cached_layer(tower, f"pd{i}", c.weight, dtype, lambda c=c: _dgrad2d_k3_layer(c.weight, dtype)))
        ctx.saved = None
        grads = []
        for c, w_, b_ in zip(convs, dW, dB):
            grads += [None if w_ is None else w_.to(c.weight.dtype), None if b_ is None else b_.to(c.bias.dtype)]
        return (None, None, None, None, *grads)


# --------------------------------------------------------------------------------------------
# generic 2-D layer nodes (Vis-MVSNet's FeatExt in train(): a residual U-Net whose graph -- shortcuts, the decoder's concat -- is left
# to torch autograd; every convolution, BatchNorm pass and their backward passes are engine launches on channels-last 16-bit maps)
# --------------------------------------------------------------------------------------------
def _wgrad_k3_sliced(dy: torch.Tensor, x: torch.Tensor, co: int, ci: int) -> torch.Tensor:
    """[co,ci,3,3] weight gradient of a k3 s1 p1 conv from dy [N,H,W,co'] and x [N,H,W,ci']; more than 64 channels on either side run
    as 64-channel slices of the same tensors (channel offsets of the weight-gradient kernel)."""
    p5, q5 = _v5(dy), _v5(x)
    if co <= 64 and ci <= 64:
        return ops.conv3d_wgrad(p5, q5, ca=co, cb=ci, stride=1)[:, :, 1]
    dw = torch.empty((co, ci, 3, 3), dtype=torch.float32, device=dy.device)
    for a0 in range(0, co, 64):
        for b0 in range(0, ci, 64):
            ca, cb = min(64, co - a0), min(64, ci - b0)
            dw[a0:a0 + ca, b0:b0 + cb] = ops.conv3d_wgrad(p5, q5, ca=ca, cb=cb, stride=1, p_coff=a0, q_coff=b0)[:, :, 1]
    return dw


_S2_TAP = {1: (0, 1), 0: (1, 0), 2: (1, 1)}     # kernel index of a k3 s2 p1 conv -> (input parity, tap of the stride-1 gradient on that parity plane)


_S2_IDX = {}


def _wgrad_k3s2(dy: torch.Tensor, x: torch.Tensor, co: int, ci: int) -> torch.Tensor:
    """k3 s2 p1: y[o] = sum_k w[k] x[2o + k - 1]; x[2o] / x[2o -+ 1] are taps of the input's parity planes, so the gradient is assembled
    from four stride-1 3x3 weight gradients (dy against x[:, a::2, b::2]) -- with ONE indexed gather (nine slice assignments were nine
    launches per layer of a host-bound step)."""
    G = torch.stack([_wgrad_k3_sliced(dy, x[:, a::2, b::2, :].contiguous(), co, ci) for a in (0, 1) for b in (0, 1)])   # [4,co,ci,3,3]
    idx = _S2_IDX.get(dy.device)
    if idx is None:
        P = [[_S2_TAP[ky][0] * 2 + _S2_TAP[kx][0] for kx in range(3)] for ky in range(3)]
        TY = [[_S2_TAP[ky][1] for kx in range(3)] for ky in range(3)]
        TX = [[_S2_TAP[kx][1] for kx in range(3)] for ky in range(3)]
        idx = _S2_IDX[dy.device] = tuple(torch.tensor(t_, dtype=torch.long, device=dy.device) for t_ in (P, TY, TX))
    P, TY, TX = idx
    return G[P, :, :, TY, TX].permute(2, 3, 0, 1).contiguous()                 # [3,3,co,ci] -> [co,ci,3,3]


class Conv2dFn(torch.autograd.Function):
    """Bias-free Conv2d (k3 s1 | k3 s2 | k1 s1 | k1 s2 | k5 s2, the reference's paddings) on channels-last 16-bit maps, forward and
    backward on the engine.  ``forward(ctx, holder, tag, dtype, stride, x [N,H,W,Ci'], w [Co,Ci,k,k])``; packed layers are cached on
    ``holder`` under ``tag`` per weight version."""

    @staticmethod
    def forward(ctx, holder, tag, dtype, stride, x, w):
        k = int(w.shape[2])
        y = ops.conv2d(x, _cached_layer(holder, f"c{tag}", w, dtype, lambda: ops.Conv2dLayer.build(w, stride=stride, dtype=dtype)))
        ctx.meta = (holder, tag, dtype, int(stride), k)
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, g):
        holder, tag, dtype, stride, k = ctx.meta
        x, w = ctx.saved_tensors
        co, ci = int(w.shape[0]), int(w.shape[1])
        g = g.contiguous()
        cpad = x.shape[3]
        if k == 3 and stride == 1:
            dw = _wgrad_k3_sliced(g, x, co, cpad)
        elif k == 3:
            dw = _wgrad_k3s2(g, x, co, cpad)
        elif k == 1:
            xs = x if stride == 1 else x[:, ::2, ::2, :].contiguous()
            dw = _wgrad_k3_sliced(g, xs, co, cpad)[:, :, 1:2, 1:2]
        else:
            dw = _wgrad2d_k5s2(g, x, co, cpad)
        dw = dw[:, :ci].to(w.dtype)
        dx = None
        if ctx.needs_input_grad[4]:
            if k == 3 and stride == 1:
                dx = ops.conv2d(g, _cached_layer(holder, f"d{tag}", w, dtype, lambda: _dgrad2d_k3_layer(w, dtype)))
            elif k == 1 and stride == 1:
                def mk():
                    w3 = torch.zeros((co, ci, 3, 3), dtype=torch.float32, device=w.device)
                    w3[:, :, 1, 1] = w.detach().float()[:, :, 0, 0]
                    return _dgrad2d_k3_layer(w3, dtype)
                dx = ops.conv2d(g, _cached_layer(holder, f"d{tag}", w, dtype, mk))
            elif k == 3:          # adjoint of a stride-2 conv = ConvTranspose2d(k3, s2, p1, op1) with the same weight: four parity sub-convs
                subs = _cached_layer(holder, f"d{tag}", w, dtype,
                                     lambda: [ops.Conv2dLayer.build(w_, stride=1, dtype=dtype) for w_ in ops.deconv2d_parity_weights(w)])
                dx = torch.empty_like(x)
                for par, sub in enumerate(subs):
                    ops.conv2d(g, sub, out=dx, parity=par)
            elif k == 1:          # strided 1x1: the gradient lands on the even pixels only (centre tap of a k3 parity-0 sub-conv)
                def mk():
                    w3 = torch.zeros((ci, co, 3, 3), dtype=torch.float32, device=w.device)
                    w3[:, :, 1, 1] = w.detach().float()[:, :, 0, 0].t()
                    return ops.Conv2dLayer.build(w3, stride=1, dtype=dtype)
                dx = torch.zeros_like(x)
                ops.conv2d(g, _cached_layer(holder, f"d{tag}", w, dtype, mk), out=dx, parity=0)
            else:
                raise NotImplementedError("pscv Conv2dFn: no data gradient for k5 s2 (the extractors' first layer reads the image)")
        return None, None, None, None, dx, dw


class Deconv2dFn(torch.autograd.Function):
    """Bias-free ConvTranspose2d(k3, s2, p1, output_padding 1) on channels-last 16-bit maps: forward = four parity sub-convolutions,
    data gradient = the k3 s2 conv with the same weight, weight gradient from the parity planes of the output gradient.
    ``forward(ctx, holder, tag, dtype, x [N,H,W,Ci], w [Ci,Co,3,3])`` -> [N,2H,2W,Co]."""

    @staticmethod
    def forward(ctx, holder, tag, dtype, x, w):
        subs = _cached_layer(holder, f"u{tag}", w, dtype,
                             lambda: [ops.Conv2dLayer.build(w_, stride=1, dtype=dtype) for w_ in ops.deconv2d_parity_weights(w)])
        N, H, W, _ = x.shape
        out = torch.empty((N, 2 * H, 2 * W, int(w.shape[1])), dtype=dtype, device=x.device)
        for par, sub in enumerate(subs):
            ops.conv2d(x, sub, out=out, parity=par)
        ctx.meta = (holder, tag, dtype)
        ctx.save_for_backward(x, w)
        return out

    @staticmethod
    def backward(ctx, g):
        holder, tag, dtype = ctx.meta
        x, w = ctx.saved_tensors
        ci, co = int(w.shape[0]), int(w.shape[1])
        g = g.contiguous()
        # dw[ci,co,k] = sum_i x[i,ci] g[2i + k - 1, co]: the stride-2 gradient with x in the role of "dy" and g in the role of "x"
        dw = _wgrad_k3s2(x, g, ci, co).to(w.dtype)
        dx = None
        if ctx.needs_input_grad[3]:
            dx = ops.conv2d(g, _cached_layer(holder, f"ud{tag}", w, dtype, lambda: ops.Conv2dLayer.build(w, stride=2, dtype=dtype)))
        return None, None, None, dx, dw


class BnAct2dFn(torch.autograd.Function):
    """``[relu](bn(y)) [+ skip]`` / ``relu(bn(y) + skip)`` of a raw conv output y [N,H,W,C] (16-bit channels-last) with batch-statistics
    BatchNorm2d over ``groups`` consecutive slices of the batch (the views: each its own statistics), forward and backward on the
    engine.  ``relu``: "pre" (before the skip add), "post" (after it: BasicBlock), None.
    ``forward(ctx, bn, groups, relu, y, skip, gamma, beta)``."""

    @staticmethod
    def forward(ctx, bn, groups, relu, y, skip, gamma, beta):
        y5 = _v5(y)
        nvox = y5.numel() // y5.shape[4] // groups
        aff = _bn_affine_grouped(bn, ops.bn_stats(y5, groups), nvox, groups)
        out = ops.bn_act(y5, aff[:, 0], aff[:, 1], relu=True if relu == "pre" else ("post" if relu == "post" else False),
                         skip=None if skip is None else _v5(skip.contiguous())).squeeze(1)
        ctx.meta = (bn, relu, nvox, skip is not None)
        ctx.save_for_backward(y, out if relu == "post" else y, aff)
        return out

    @staticmethod
    def backward(ctx, g):
        bn, relu, nvox, has_skip = ctx.meta
        y, out, aff = ctx.saved_tensors
        g = g.contiguous()
        dpre = ops.relu_bwd(_v5(g), _v5(out)).squeeze(1) if relu == "post" else g
        s = ops.bn_bwd_reduce(_v5(dpre), _v5(y), aff[:, 0], aff[:, 1], relu=relu == "pre")
        cf = ops.bn_bwd_coeffs(s, aff[:, 2], aff[:, 3], bn.weight, nvox)
        if not bn.training:
            cf[:, 1:3].zero_()
        dy = ops.bn_bwd_apply(_v5(dpre), _v5(y), aff[:, 0], aff[:, 1], cf[:, 0], cf[:, 1], cf[:, 2], relu=relu == "pre").squeeze(1)
        dgb = cf[:, 3:5].sum(0)
        return (None, None, None, dy, dpre if has_skip else None,
                dgb[0].to(bn.weight.dtype) if bn.weight is not None else None, dgb[1].to(bn.bias.dtype) if bn.bias is not None else None)


# --------------------------------------------------------------------------------------------
# unsupervised photometric loss (models/trainer.py:209-278, utils/ssimLoss.py)
# --------------------------------------------------------------------------------------------
class PhotoWarpFn(torch.autograd.Function):
    """depth [B,h,w] -> (warped sources [B,S,C,h,w], mask [B,S,h,w]); the gradient flows to the DEPTH through the sampling
    position, as in the reference (the images are data).  Forward and backward are one HIP launch each."""

    @staticmethod
    def forward(ctx, depth, src_imgs, inv_ref, proj_src):
        o = ops.photo_warp(src_imgs, depth.detach(), inv_ref, proj_src)
        ctx.save_for_backward(depth.detach(), src_imgs, inv_ref, proj_src)
        ctx.mark_non_differentiable(o["mask"])
        return o["warped"], o["mask"]

    @staticmethod
    def backward(ctx, grad_warped, _grad_mask):
        depth, src_imgs, inv_ref, proj_src = ctx.saved_tensors
        return ops.photo_warp_bwd(src_imgs, depth, inv_ref, proj_src, grad_warped.contiguous()), None, None, None


class SSIMFn(torch.autograd.Function):
    """1 - SSIM(img1, img2) per channel; differentiable in img2 (the warped image), img1 is data."""

    @staticmethod
    def forward(ctx, img1, img2):
        if img1.requires_grad:
            raise NotImplementedError("pscv SSIM: the first image is treated as data (the reference image of the loss)")
        ctx.save_for_backward(img1, img2.detach())
        return ops.ssim(img1, img2.detach())

    @staticmethod
    def backward(ctx, grad_out):
        img1, img2 = ctx.saved_tensors
        return None, ops.ssim_bwd(img1, img2, grad_out.contiguous())
