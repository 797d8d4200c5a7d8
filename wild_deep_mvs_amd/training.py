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

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

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
        ref_cl = ops.to_channels_last(ref.detach(), dtype)
        srcs_cl = [ops.to_channels_last(s.detach(), dtype) for s in srcs]
        tval = float(temp.detach().float().item()) if temp is not None else 0.0
        out = ops.warp_cost(ref_cl, srcs_cl, cams, depth, geom=geom, cost=cost, temp=tval, out_dtype=dtype)
        ctx.save_for_backward(cams, depth, ref_cl, *srcs_cl)
        ctx.meta = (geom, cost, tval, temp is not None, ref.dtype, [s.dtype for s in srcs])
        return out

    @staticmethod
    def backward(ctx, g):
        cams, depth, ref_cl, *srcs_cl = ctx.saved_tensors
        geom, cost, tval, has_temp, ref_dt, src_dts = ctx.meta
        g = g.contiguous()
        dref, dsrcs, dtemp = ops.warp_cost_bwd(ref_cl, srcs_cl, cams, depth, g, geom=geom, cost=cost, temp=tval,
                                               want_dtemp=has_temp)
        gref = dref.permute(0, 3, 1, 2).to(ref_dt)
        gsrcs = [d.permute(0, 3, 1, 2).to(dt) for d, dt in zip(dsrcs, src_dts)]
        return (None, None, None, None, None, dtemp if has_temp else None, gref, *gsrcs)


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
    """Batch statistics -> (scale, bias, mean, invstd); updates the running statistics like nn.BatchNorm3d.train()."""
    bn = b.bn
    mean = sums[0] / nvox
    var = (sums[1] / nvox - mean * mean).clamp_min_(0.0)
    invstd = torch.rsqrt(var + bn.eps)
    gamma = bn.weight.detach().float() if bn.weight is not None else torch.ones_like(mean)
    beta = bn.bias.detach().float() if bn.bias is not None else torch.zeros_like(mean)
    scale = (gamma * invstd).contiguous()
    bias = (beta - mean * scale).contiguous()
    if bn.track_running_stats and bn.running_mean is not None:
        with torch.no_grad():
            if bn.num_batches_tracked is not None:
                bn.num_batches_tracked += 1
            m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked.item())
            bn.running_mean.mul_(1.0 - m).add_(mean.to(bn.running_mean.dtype), alpha=m)
            unbiased = var * (nvox / max(nvox - 1, 1))
            bn.running_var.mul_(1.0 - m).add_(unbiased.to(bn.running_var.dtype), alpha=m)
    return scale, bias, mean, invstd


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
            s1 = s[0]
            s2 = invstd * (s[1] - mean * s[0])                      # sum dz * xhat
            gamma = b.bn.weight.detach().float()
            k = gamma * invstd
            ca = k.contiguous()
            cb = (-k * invstd * s2 / nvox).contiguous()
            cc = (-k * s1 / nvox + k * invstd * mean * s2 / nvox).contiguous()
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
