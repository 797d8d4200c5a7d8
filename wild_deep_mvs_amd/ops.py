"""Tensor-level host wrappers over the pscv C ABI (include/pscv.h).

PyTorch is plumbing here: it owns device memory and the stream; every op below hands raw device
pointers to ``libpscv.so`` on ``torch.cuda.current_stream()``.  Nothing in this module computes the
hot path with torch ops, and every entry point raises if the tensors are not on a HIP device.

Layouts: feature maps ``[B,h,w,C]`` and volumes ``[B,D,h,w,C]`` (channels-last), bf16 or fp32 storage.
"""
from __future__ import annotations

import ctypes as C
import weakref
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib as L

_TORCH2PSCV = {torch.float32: L.F32, torch.bfloat16: L.BF16, torch.float16: L.F16}
HALF_DTYPES = (torch.bfloat16, torch.float16)


def _dt(t: torch.Tensor) -> int:
    try:
        return _TORCH2PSCV[t.dtype]
    except KeyError:
        raise TypeError(f"pscv: unsupported dtype {t.dtype} (float32 / bfloat16 / float16 only)")


def _dev(*ts: Optional[torch.Tensor]):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("pscv: the plane-sweep engine runs on MI355X only; got a CPU tensor "
                               "(there is no CPU / PyTorch fallback for the hot path)")
        if not t.is_contiguous():
            raise ValueError("pscv: tensors handed to the C ABI must be contiguous")


def _p(t: Optional[torch.Tensor]):
    """Device address for a ``c_void_p`` parameter (every entry point declares its argtypes): the plain integer / None -- ctypes
    converts it at the call; building a ``c_void_p`` object per pointer cost ~2 us per launch."""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """Raw handle of the current HIP stream.  ``torch.cuda.current_stream()`` builds a Stream object through several Python layers
    (~4-7 us per call -- more than the ctypes launch itself, and a forward of Vis-MVSNet makes ~300 launches); the C-level accessor
    is the same query without the object."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


class EventTimer:
    """Per-kernel HIP-event timing on the launch stream (used by bench.py for the roofline figures).

    ``with EventTimer() as tm: ...`` brackets every pscv launch with a pair of events recorded on the
    stream the kernel is launched on; ``tm.summary()`` (after a device sync) returns
    ``{name: (launches, total_ms)}``."""

    def __init__(self):
        self.records = []
        self.costs = []

    def __enter__(self):
        global _timer
        self._prev, _timer = _timer, self
        return self

    def __exit__(self, *exc):
        global _timer
        _timer = self._prev
        return False

    def launch(self, name, fn, cost=None):
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        rc = fn()
        e1.record(st)
        self.records.append((name, e0, e1))
        if cost is not None:
            self.costs.append((name,) + tuple(cost()))
        return rc

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1 in self.records:
            n, ms = out.get(name, (0, 0.0))
            out[name] = (n + 1, ms + e0.elapsed_time(e1))
        return out

    def detail(self):
        """{name: dict(launches, ms, bytes, flops)}: ``bytes`` / ``flops`` are the ALGORITHMIC HBM bytes (operands read once,
        results written once) and floating-point operations the launches of that name asked for (conv3d / conv2d / warp_cost
        report them; 0 for the others)."""
        out = {k: dict(launches=n, ms=ms, bytes=0.0, flops=0.0) for k, (n, ms) in self.summary().items()}
        for name, b, f in self.costs:
            out[name]["bytes"] += b
            out[name]["flops"] += f
        return out


_timer: Optional[EventTimer] = None


def _launch(name, fn, cost=None):
    """Run a C-ABI launch; under an EventTimer it is bracketed by HIP events and ``cost()`` -> (algorithmic bytes, flops) is
    recorded with it (evaluated only then)."""
    return _timer.launch(name, fn, cost) if _timer is not None else fn()


# --------------------------------------------------------------------------------------------
# layout helpers (plumbing)
# --------------------------------------------------------------------------------------------
def to_channels_last(x: torch.Tensor, dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    """[B,C,h,w] -> [B,h,w,C] or [B,C,D,h,w] -> [B,D,h,w,C], contiguous, in the storage dtype."""
    perm = (0, 2, 3, 1) if x.dim() == 4 else (0, 2, 3, 4, 1)
    return x.permute(*perm).to(dtype).contiguous()


def to_channels_first(x: torch.Tensor) -> torch.Tensor:
    """View a channels-last pscv tensor with the reference's NCHW / NCDHW index order (no copy)."""
    perm = (0, 3, 1, 2) if x.dim() == 4 else (0, 4, 1, 2, 3)
    return x.permute(*perm)


# --------------------------------------------------------------------------------------------
# cameras (A0; stays torch, fp64 closed forms, no LAPACK call on the device)
# --------------------------------------------------------------------------------------------
def inv3x3(m: torch.Tensor) -> torch.Tensor:
    """Batched adjugate inverse of [...,3,3]."""
    a, b, c = m[..., 0, 0], m[..., 0, 1], m[..., 0, 2]
    d, e, f = m[..., 1, 0], m[..., 1, 1], m[..., 1, 2]
    g, h, i = m[..., 2, 0], m[..., 2, 1], m[..., 2, 2]
    co = torch.stack([e * i - f * h, c * h - b * i, b * f - c * e,
                      f * g - d * i, a * i - c * g, c * d - a * f,
                      d * h - e * g, b * g - a * h, a * e - b * d], dim=-1)
    det = a * co[..., 0] + b * co[..., 3] + c * co[..., 6]
    return (co / det.unsqueeze(-1)).reshape(m.shape)


def proj_cams(src_projs: Sequence[torch.Tensor], ref_proj: torch.Tensor) -> torch.Tensor:
    """PROJ-geometry camera block [n_src,B,18]: rot[9], trans[3] of ``P_src P_ref^-1``
    (reference models/MVSNet/module.py:128-130).  Projection matrices are [B,4,4] with last row (0,0,0,1)."""
    Pr = ref_proj.double()
    Ar_inv = inv3x3(Pr[:, :3, :3])
    br = Pr[:, :3, 3:4]
    out = []
    for Ps in src_projs:
        Ps = Ps.double()
        rot = Ps[:, :3, :3] @ Ar_inv
        trans = Ps[:, :3, 3:4] - rot @ br
        blk = torch.cat([rot.reshape(-1, 9), trans.reshape(-1, 3), torch.zeros_like(rot.reshape(-1, 9)[:, :6])], dim=1)
        out.append(blk)
    return torch.stack(out).to(torch.float32).contiguous()


def proj_cams_device(proj: torch.Tensor, reference_frame: int = 0) -> torch.Tensor:
    """Same block as ``proj_cams`` for all source views of ``proj`` [B,V,4,4] in ONE HIP launch
    (pscv_proj_cams): [V-1,B,18], sources in view order with the reference skipped."""
    _dev(proj)
    if proj.dim() != 4 or proj.shape[2:] != (4, 4) or proj.dtype != torch.float32:
        raise ValueError("pscv.proj_cams_device: proj must be fp32 [B,V,4,4]")
    B, V = proj.shape[:2]
    cams = torch.empty((V - 1, B, L.CAM_FLOATS), dtype=torch.float32, device=proj.device)
    rc = _launch("proj_cams", lambda: L.lib().pscv_proj_cams(_p(proj), B, V, int(reference_frame), _p(cams), _stream()))
    L.check(rc, "pscv_proj_cams")
    return cams


def batch_views(ts: Sequence[torch.Tensor]) -> torch.Tensor:
    """``torch.cat(ts, 0)`` -- as a VIEW when the tensors are back-to-back slices of one buffer (the views of an image batch
    [1,V,3,H,W] picked with ``imgs[:, i]``): the extractors batch all views without a copy."""
    v = _consecutive_views(ts)
    if v is None:
        return torch.cat(list(ts), 0)
    return v.reshape((len(ts) * ts[0].shape[0],) + tuple(ts[0].shape[1:]))


def _consecutive_views(ts: Sequence[torch.Tensor]) -> Optional[torch.Tensor]:
    """[n, *shape] view over ``ts`` when they are equally shaped contiguous tensors lying back to back in one storage (e.g.
    ``base[1:].unbind(0)``): no copy.  None otherwise."""
    ts = list(ts)
    t0 = ts[0]
    if not all(t.is_contiguous() and t.shape == t0.shape and t.dtype == t0.dtype and t.device == t0.device
               and t.untyped_storage().data_ptr() == t0.untyped_storage().data_ptr() for t in ts):
        return None
    step = t0.numel()
    if step == 0 or any(t.storage_offset() != t0.storage_offset() + i * step for i, t in enumerate(ts)):
        return None
    return torch.as_strided(t0, (len(ts),) + tuple(t0.shape), (step,) + tuple(t0.stride()))


def homog_cams_device(ref_cam: torch.Tensor, src_cams: Sequence[torch.Tensor], scale: float) -> torch.Tensor:
    """HOMOG-geometry camera blocks [n_src,B,18] (A[9], Bm[9]) from Vis-style cam arrays [B,2,4,4] in one HIP
    launch (pscv_homog_cams); ``scale`` = 1 / s_scale is applied to the intrinsics like scale_camera."""
    src = _consecutive_views(src_cams)
    if src is None:
        src = torch.stack(list(src_cams))
    src = src.to(torch.float32).contiguous()
    ref = ref_cam.to(torch.float32).contiguous()
    _dev(ref, src)
    n, B = src.shape[:2]
    cams = torch.empty((n, B, L.CAM_FLOATS), dtype=torch.float32, device=ref.device)
    rc = _launch("homog_cams", lambda: L.lib().pscv_homog_cams(_p(ref), _p(src), B, n, float(scale), _p(cams), _stream()))
    L.check(rc, "pscv_homog_cams")
    return cams


def fuse_pairs(interms: Sequence[torch.Tensor], uncerts: Sequence[torch.Tensor], *, normalise: bool = True,
               want_wsum: bool = False):
    """interms n x [B,D,h,w,8] (16-bit), uncerts n x [B,h,w] fp32 -> fused [B,D,h,w,8] (same dtype, normalised)
    or fp32 partial sums (normalise=False); optionally also the weight sum [B,h,w]."""
    interms, uncerts = list(interms), list(uncerts)
    _dev(*interms, *uncerts)
    B, D, h, w, c = interms[0].shape
    if c != 8 or any(t.shape != interms[0].shape or t.dtype != interms[0].dtype for t in interms):
        raise ValueError("pscv.fuse_pairs: pair volumes must share shape [B,D,h,w,8] and dtype")
    if any(u.dtype != torch.float32 or tuple(u.shape) != (B, h, w) for u in uncerts) or len(uncerts) != len(interms):
        raise ValueError("pscv.fuse_pairs: uncertainty maps must be fp32 [B,h,w], one per pair volume")
    n = len(interms)
    out = torch.empty((B, D, h, w, 8), dtype=interms[0].dtype if normalise else torch.float32, device=interms[0].device)
    wsum = torch.empty((B, h, w), dtype=torch.float32, device=out.device) if want_wsum else None
    ip = (C.c_void_p * n)(*[t.data_ptr() for t in interms])
    up = (C.c_void_p * n)(*[t.data_ptr() for t in uncerts])
    rc = _launch("fuse_pairs", lambda: L.lib().pscv_fuse_pairs(ip, up, n, _dt(interms[0]), _p(out), _p(wsum), int(normalise),
                                                               B, D, h, w, _stream()))
    L.check(rc, "pscv_fuse_pairs")
    return (out, wsum) if want_wsum else out


def fuse_finish(partial: torch.Tensor, wsum: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """partial fp32 [B,D,h,w,8] (all-reduced sum of w*interm), wsum fp32 [B,h,w] -> normalised 16-bit volume."""
    _dev(partial, wsum)
    B, D, h, w, c = partial.shape
    if c != 8 or partial.dtype != torch.float32 or tuple(wsum.shape) != (B, h, w) or wsum.dtype != torch.float32:
        raise ValueError("pscv.fuse_finish: partial fp32 [B,D,h,w,8] and wsum fp32 [B,h,w] expected")
    out = torch.empty((B, D, h, w, 8), dtype=dtype, device=partial.device)
    rc = _launch("fuse_finish", lambda: L.lib().pscv_fuse_finish(_p(partial), _p(wsum), _TORCH2PSCV[dtype], _p(out), B, D, h, w, _stream()))
    L.check(rc, "pscv_fuse_finish")
    return out


# --------------------------------------------------------------------------------------------
# 2-D feature extractor layers (the step before the path)
# --------------------------------------------------------------------------------------------
def pack_conv2d_weights(weight: torch.Tensor, c_in_padded: int, dtype: torch.dtype = torch.float16) -> np.ndarray:
    """Host-side repack of a Conv2d weight [Co,Ci,k,k] (k = 3 or 5) into the MFMA fragment order of pscv_conv2d for
    an input that carries ``c_in_padded`` >= Ci channels (uint16 bit patterns of ``dtype``)."""
    w = np.ascontiguousarray(weight.detach().to("cpu", torch.float32).numpy())
    if w.ndim != 4 or w.shape[2] != w.shape[3]:
        raise ValueError(f"pscv: conv2d weights must be [Co,Ci,k,k], got {w.shape}")
    c_out, c_in, ks = w.shape[0], w.shape[1], w.shape[2]
    lib = L.lib()
    code = _TORCH2PSCV.get(dtype, -1)
    n = lib.pscv_pack_conv2d_weights(None, c_in, c_in_padded, c_out, ks, code, None)
    if n < 0:
        L.check(int(n), "pscv_pack_conv2d_weights")
    packed = np.empty(n, dtype=np.uint16)
    n2 = lib.pscv_pack_conv2d_weights(w.ctypes.data_as(C.c_void_p), c_in, c_in_padded, c_out, ks, code, packed.ctypes.data_as(C.c_void_p))
    if n2 != n:
        L.check(-1 if n2 >= 0 else int(n2), "pscv_pack_conv2d_weights")
    return packed


@dataclass
class Conv2dLayer:
    """One 2-D layer ready for the engine: packed 16-bit weights + fp32 epilogue vectors, on device."""
    packed: torch.Tensor
    dtype: torch.dtype
    c_in: int        # channels of the input map (padded count)
    c_out: int
    ks: int
    stride: int
    neg_slope: float   # activation max(v, neg_slope * v): 0 = ReLU, 0.1 = LeakyReLU(0.1), 1 = none
    scale: Optional[torch.Tensor] = None
    bias: Optional[torch.Tensor] = None

    @staticmethod
    def build(weight: torch.Tensor, *, stride: int, device=None, bn: Optional[Sequence[torch.Tensor]] = None,
              bn_eps: float = 1e-5, conv_bias: Optional[torch.Tensor] = None, relu: bool = False,
              leaky: Optional[float] = None, dtype: torch.dtype = torch.float16, adjoint: bool = False) -> "Conv2dLayer":
        """``bn`` = (gamma, beta, running_mean, running_var) folds an eval-mode BatchNorm2d into the epilogue;
        ``relu`` / ``leaky`` (negative slope) select the fused activation.  ``adjoint``: ``weight`` is a stride-1 FORWARD layer's
        [Co,Ci,k,k] and the result is its adjoint (Co -> Ci, taps flipped), packed straight from that tensor on the device."""
        device = device if device is not None else weight.device
        if adjoint:
            if not (weight.is_cuda and torch.device(device) == weight.device):
                return Conv2dLayer.build(weight.detach().float().flip(2, 3).transpose(0, 1).contiguous(), stride=stride, device=device, bn=bn,
                                         bn_eps=bn_eps, conv_bias=conv_bias, relu=relu, leaky=leaky, dtype=dtype)
            c_in, c_out, ks = int(weight.shape[0]), int(weight.shape[1]), int(weight.shape[2])
        else:
            c_out, c_in, ks = int(weight.shape[0]), int(weight.shape[1]), int(weight.shape[2])
        c_pad = (c_in + 7) // 8 * 8
        if weight.is_cuda and torch.device(device) == weight.device and weight.dim() == 4 and weight.shape[2] == weight.shape[3]:
            # device-side packing (same bits): no device -> host copy, no stream synchronisation per layer
            wd = weight.detach().to(torch.float32).contiguous()
            n = L.lib().pscv_pack_conv2d_weights(None, c_in, c_pad, c_out, ks, _TORCH2PSCV.get(dtype, -1), None)
            if n < 0:
                L.check(int(n), "pscv_pack_conv2d_weights")
            packed = torch.empty(int(n), dtype=torch.int16, device=weight.device)
            with torch.cuda.device(weight.device):
                L.check(L.lib().pscv_pack_conv2d_weights_device_ex(_p(wd), c_in, c_pad, c_out, ks, _TORCH2PSCV[dtype], int(adjoint), _p(packed),
                                                                   _stream()), "pscv_pack_conv2d_weights_device")
        else:
            packed = torch.from_numpy(pack_conv2d_weights(weight, c_pad, dtype).view(np.int16)).to(device)
        scale = bias = None
        if bn is not None:
            gamma, beta, mean, var = [t.detach().to(device, torch.float32) for t in bn]
            scale = gamma / torch.sqrt(var + bn_eps)
            bias = beta - mean * scale
            if conv_bias is not None:
                bias = bias + scale * conv_bias.detach().to(device, torch.float32)
            scale, bias = scale.contiguous(), bias.contiguous()
        elif conv_bias is not None:
            bias = conv_bias.detach().to(device, torch.float32).contiguous()
        slope = float(leaky) if leaky is not None else (0.0 if relu else 1.0)
        return Conv2dLayer(packed, dtype, c_pad, c_out, ks, int(stride), slope, scale, bias)


def conv2d_out_hw(layer: "Conv2dLayer", H: int, W: int, parity: int = -1):
    if layer.ks == 2 or parity >= 0:       # parity sub-convolution of a stride-2 transposed conv: full output map is 2x
        return 2 * H, 2 * W
    pad = layer.ks // 2
    return (H + 2 * pad - layer.ks) // layer.stride + 1, (W + 2 * pad - layer.ks) // layer.stride + 1


def conv2d(x: torch.Tensor, layer: Conv2dLayer, *, out_dtype: Optional[torch.dtype] = None, skip: Optional[torch.Tensor] = None,
           skip_coff: int = 0, out: Optional[torch.Tensor] = None, out_coff: int = 0, parity: int = -1) -> torch.Tensor:
    """x [B,H,W,C] in the layer's 16-bit format -> [B,Ho,Wo,c_out] in the same format or fp32 (pscv_conv2d_ex).
    ``skip`` is added before the activation; ``out`` / ``out_coff`` write a channel slice of a wider map; ``parity``
    (0..3) selects the output parity of a ks = 2 sub-convolution of a transposed conv (writes pixels (2i+py, 2j+px))."""
    _dev(x, layer.packed, skip, out)
    if x.dtype != layer.dtype or x.dim() != 4 or x.shape[3] != layer.c_in:
        raise TypeError(f"pscv.conv2d: input must be a {layer.dtype} [B,H,W,{layer.c_in}] map, got {x.dtype} {tuple(x.shape)}")
    out_dtype = layer.dtype if out_dtype is None else out_dtype
    B, H, W, _ = x.shape
    Ho, Wo = conv2d_out_hw(layer, H, W, parity)
    if out is None:
        out = torch.empty((B, Ho, Wo, layer.c_out), dtype=out_dtype, device=x.device)
    if tuple(out.shape[:3]) != (B, Ho, Wo):
        raise ValueError(f"pscv.conv2d: out has shape {tuple(out.shape)}, expected [B,{Ho},{Wo},*]")
    if skip is not None and (skip.dtype != layer.dtype or tuple(skip.shape[:3]) != (B, Ho, Wo)):
        raise ValueError("pscv.conv2d: skip must have the layer's dtype and the output's spatial shape")
    rc = _launch(f"conv2d[{layer.c_in}->{layer.c_out},k{layer.ks}s{layer.stride}]", lambda: L.lib().pscv_conv2d_ex(
        _p(x), _dt(x), _p(layer.packed), _p(layer.scale), _p(layer.bias), _p(skip), 0 if skip is None else skip.shape[3], skip_coff,
        _p(out), out.shape[3], out_coff, _dt(out), B, H, W, layer.c_in, layer.c_out, layer.ks, layer.stride, int(parity),
        float(layer.neg_slope), _stream()),
        # a parity launch (one of the four k2 sub-convolutions of a stride-2 transposed conv) writes ONE output pixel per input pixel --
        # a quarter of the 2H x 2W map it addresses -- so its output-side bytes and FLOPs count B*H*W pixels, not B*Ho*Wo
        cost=lambda: _conv_cost(B * H * W, B * H * W if parity >= 0 else B * Ho * Wo, layer.c_in, layer.c_out, layer.ks * layer.ks, False,
                                out.element_size(), skip is not None))
    L.check(rc, "pscv_conv2d")
    return out


_PARITY_K = {}


def deconv2d_parity_weights(weight: torch.Tensor):
    """ConvTranspose2d(k3, s2, p1, op1) weight [Ci,Co,3,3] -> four Conv2d weights [Co,Ci,2,2], index 2*py+px: output pixel
    (2i+py, 2j+px) = sum over taps (ty, tx) of W[py,px][:, :, ty, tx] x[i+ty, j+tx].  Along a dim, parity 0 takes input i with
    kernel index 1; parity 1 takes input i with kernel index 2 and input i+1 with kernel index 0."""
    w = weight.detach().float()
    ci, co = w.shape[:2]
    # kernel index per (parity, tap): parity 0 -> (1, none), parity 1 -> (2, 0); "none" reads a zero plane appended at index 3.
    # One indexed gather for all four sub-weights (16 slice assignments before: a training step rebuilds these per layer)
    wp = torch.nn.functional.pad(w, (0, 1, 0, 1))                                   # [ci,co,4,4], row / column 3 = zeros
    k = _PARITY_K.get(w.device)
    if k is None:
        k = _PARITY_K[w.device] = torch.tensor([[1, 3], [2, 0]], dtype=torch.long, device=w.device)
    W = wp[:, :, k[:, :, None, None], k[None, None, :, :]]                          # [ci,co,py,ty,px,tx]
    return [W[:, :, py, :, px, :].transpose(0, 1).contiguous() for py in (0, 1) for px in (0, 1)]


def image_to_channels_last8(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """[B,3,H,W] image -> [B,H,W,8] in the storage dtype, channels 3-7 zero (the first layer's padded input): one HIP launch
    (pscv_image_prep) for fp32 images on the GPU, torch ops otherwise."""
    B, c, H, W = x.shape
    if x.is_cuda and x.dtype == torch.float32 and c <= 8 and B <= 65535:
        x = x.contiguous()
        out = torch.empty((B, H, W, 8), dtype=dtype, device=x.device)
        rc = _launch("image_prep", lambda: L.lib().pscv_image_prep(_p(x), B, c, H, W, _dt(out), _p(out), None, None, _stream()))
        L.check(rc, "pscv_image_prep")
        return out
    out = torch.zeros((B, H, W, 8), dtype=dtype, device=x.device)
    out[..., :c] = x.permute(0, 2, 3, 1)
    return out


def image_pyramid_cl8(x: torch.Tensor, scales: int, dtype: torch.dtype) -> List[torch.Tensor]:
    """CVP's image pyramid in the extractor layout, finest first: level l+1 = F.interpolate(level l, scale_factor=0.5,
    mode='bilinear') (net.py:34-47), every level as [B,H_l,W_l,8] 16-bit pixels.  One launch per level reads the fp32 image once
    and writes its half-resolution fp32 image and that one's 8-channel form (pscv_image_prep); odd widths fall back to
    F.interpolate for that step."""
    B, c, H, W = x.shape
    if not (x.is_cuda and x.dtype == torch.float32 and c <= 8 and B <= 65535):
        raise TypeError("pscv.image_pyramid_cl8: fp32 [B,C<=8,H,W] images on the GPU expected")
    x = x.contiguous()
    levels = [image_to_channels_last8(x, dtype)]
    for l in range(1, scales):
        B, c, H, W = x.shape
        if W % 2 or H < 2:
            x = torch.nn.functional.interpolate(x, scale_factor=0.5, mode="bilinear", align_corners=None)
            levels.append(image_to_channels_last8(x, dtype))
            continue
        last = l == scales - 1
        half = None if last else torch.empty((B, c, H // 2, W // 2), dtype=torch.float32, device=x.device)
        cl = torch.empty((B, H // 2, W // 2, 8), dtype=dtype, device=x.device)
        rc = _launch("image_prep", lambda: L.lib().pscv_image_prep(_p(x), B, c, H, W, _dt(cl), None, _p(half), _p(cl), _stream()))
        L.check(rc, "pscv_image_prep")
        levels.append(cl)
        x = half
    return levels


# --------------------------------------------------------------------------------------------
# geometric-consistency filter (the step after the path)
# --------------------------------------------------------------------------------------------
def geo_filter_cams(K: torch.Tensor, R: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """[V,3,3], [V,3,3], [V,3,1] (view 0 = reference) -> camera blocks [V,30] fp32 (K, K^-1, R, t) for
    ``geo_filter``.  The inverse is ``torch.inverse`` in fp32, what the reference uses (utils_3D.py:131,155)."""
    K = K.to(torch.float32)
    blocks = torch.cat((K.reshape(-1, 9), torch.inverse(K).reshape(-1, 9), R.to(torch.float32).reshape(-1, 9),
                        t.to(torch.float32).reshape(-1, 3)), dim=1)
    return blocks.contiguous()


def geo_filter(depth: torch.Tensor, src_depth: Sequence[torch.Tensor], cams: torch.Tensor, *, max_reproj_error: float = 1.0,
               depth_threshold: float = 0.01, min_tri_angle: float = 1.0, num_consistent: int = 3,
               want_counts: bool = False):
    """depth [h,w] fp32, src_depth N x [h_i,w_i] fp32, cams [N+1,30] (``geo_filter_cams``), all on the GPU ->
    (mask_depth, mask_disp, geo_mask) bool [h,w] (and int32 counts [3,h,w] with ``want_counts``): the masks of
    evaluation/filtering.py:73-83 in one launch (pscv_geo_filter)."""
    src_depth = [s.to(torch.float32).contiguous() for s in src_depth]
    depth = depth.to(torch.float32).contiguous()
    cams = cams.to(torch.float32).contiguous()
    _dev(depth, cams, *src_depth)
    n = len(src_depth)
    if n < 1 or n > L.GEO_MAX_SRC or tuple(cams.shape) != (n + 1, L.GEO_CAM_FLOATS) or depth.dim() != 2:
        raise ValueError(f"pscv.geo_filter: depth [h,w], 1..{L.GEO_MAX_SRC} source maps and cams [N+1,{L.GEO_CAM_FLOATS}] expected")
    h, w = depth.shape
    masks = torch.empty((3, h, w), dtype=torch.uint8, device=depth.device)
    counts = torch.empty((3, h, w), dtype=torch.int32, device=depth.device) if want_counts else None
    ptrs = (C.c_void_p * n)(*[s.data_ptr() for s in src_depth])
    hw = (C.c_int * (2 * n))(*[v for s in src_depth for v in s.shape])
    rc = _launch("geo_filter", lambda: L.lib().pscv_geo_filter(
        _p(depth), ptrs, hw, n, _p(cams), h, w, float(max_reproj_error), float(depth_threshold), float(min_tri_angle),
        int(num_consistent), _p(masks[0]), _p(masks[1]), _p(masks[2]), _p(counts), _stream()))
    L.check(rc, "pscv_geo_filter")
    out = (masks[0].bool(), masks[1].bool(), masks[2].bool())
    return out + (counts,) if want_counts else out


# --------------------------------------------------------------------------------------------
# fused warp + cost
# --------------------------------------------------------------------------------------------
def warp_cost(ref: Optional[torch.Tensor], srcs: Sequence[torch.Tensor], cams: torch.Tensor, depth: torch.Tensor, *,
              geom: int = L.GEOM_PROJ, cost: int = L.COST_VARIANCE, temp: float = 0.0,
              ref_hw: Optional[Sequence[int]] = None, out_dtype: torch.dtype = torch.bfloat16,
              out: Optional[torch.Tensor] = None, ref_y0: int = 0) -> torch.Tensor:
    """ref [B,h,w,C] (None for WARP_ONLY), srcs n x [B,hs,ws,C], cams [n,B,18] fp32,
    depth [B,D] or [B,D,h,w] fp32  ->  cost volume [B,D,h,w,C]
    (GROUPCORR: [n,B,D,h,w,C/4]; WARP_ONLY: [n,B,D,h,w,C]).  ``ref_y0`` > 0: ``ref`` / per-pixel ``depth`` / the result are rows
    [ref_y0, ref_y0 + h) of a larger reference grid whose cameras ``cams`` are (pscv_warp_cost_rows: bit-identical to those rows of
    the whole-image launch)."""
    srcs = list(srcs)
    _dev(ref, cams, depth, *srcs)
    B, hs, ws, Cc = srcs[0].shape
    for s in srcs:
        if s.shape != srcs[0].shape or s.dtype != srcs[0].dtype:
            raise ValueError("pscv.warp_cost: all source feature maps must share shape and dtype")
    if ref is not None:
        h, w = ref.shape[1:3]
        if ref.dtype != srcs[0].dtype or ref.shape[0] != B or ref.shape[3] != Cc:
            raise ValueError("pscv.warp_cost: ref / src feature mismatch")
    else:
        h, w = (hs, ws) if ref_hw is None else (int(ref_hw[0]), int(ref_hw[1]))
    n = len(srcs)
    if cams.shape != (n, B, L.CAM_FLOATS) or cams.dtype != torch.float32:
        raise ValueError(f"pscv.warp_cost: cams must be fp32 [{n},{B},{L.CAM_FLOATS}], got {tuple(cams.shape)}")
    if depth.dtype != torch.float32:
        raise ValueError("pscv.warp_cost: depth planes must be fp32")
    D = depth.shape[1]
    per_pixel = depth.dim() == 4
    if per_pixel and tuple(depth.shape) != (B, D, h, w):
        raise ValueError("pscv.warp_cost: per-pixel depth must be [B,D,h,w]")
    if not per_pixel and depth.dim() != 2:
        raise ValueError("pscv.warp_cost: depth must be [B,D] or [B,D,h,w]")
    bstride = depth.stride(0)
    if cost == L.COST_GROUPCORR:
        shape = (n, B, D, h, w, Cc // 4)
    elif cost == L.COST_WARP_ONLY:
        shape = (n, B, D, h, w, Cc)
    elif cost == L.COST_VARIANCE_PARTIAL:
        shape, out_dtype = (2, B, D, h, w, Cc), torch.float32     # (sum f, sum f^2) over the given views
    else:
        shape = (B, D, h, w, Cc)
    if out is None:
        out = torch.empty(shape, dtype=out_dtype, device=srcs[0].device)
    elif tuple(out.shape) != shape or not out.is_contiguous():
        raise ValueError("pscv.warp_cost: bad `out` tensor")
    ptrs = (C.c_void_p * n)(*[s.data_ptr() for s in srcs])
    rc = _launch(f"warp_cost[{cost}]", lambda: L.lib().pscv_warp_cost_rows(
        _p(ref), ptrs, n, _p(cams), _p(depth), bstride, int(per_pixel), geom, cost, float(temp), _p(out), B, Cc, h, w,
        hs, ws, D, _dt(srcs[0]), _dt(out), int(ref_y0), _stream()),
        # feature maps once + the volume once; per (voxel, view, channel) 4 blend FMAs + the cost statistic (2 FMAs)
        cost=lambda: (float((n + (ref is not None)) * B * hs * ws * Cc * srcs[0].element_size() + out.numel() * out.element_size()),
                      2.0 * 6 * n * B * D * h * w * Cc))
    L.check(rc, "pscv_warp_cost")
    return out


def variance_finish(sums: torch.Tensor, n_views: int, *, cost: int = L.COST_VARIANCE, dtype: torch.dtype = torch.float16) -> torch.Tensor:
    """(all-reduced) partial sums fp32 [2,B,D,h,w,C] of ``warp_cost(cost=COST_VARIANCE_PARTIAL)`` -> variance volume
    [B,D,h,w,C] in ``dtype`` (pscv_variance_finish)."""
    _dev(sums)
    if sums.dtype != torch.float32 or sums.dim() != 6 or sums.shape[0] != 2 or not sums.is_contiguous():
        raise ValueError("pscv.variance_finish: fp32 [2,B,D,h,w,C] partial sums expected")
    out = torch.empty(tuple(sums.shape[1:]), dtype=dtype, device=sums.device)
    rc = _launch("variance_finish", lambda: L.lib().pscv_variance_finish(_p(sums), out.numel(), int(n_views), cost, _TORCH2PSCV[dtype],
                                                                        _p(out), _stream()))
    L.check(rc, "pscv_variance_finish")
    return out


# --------------------------------------------------------------------------------------------
# conv3d
# --------------------------------------------------------------------------------------------
def pack_conv3d_weights(weight: torch.Tensor, kind: int, transposed: bool,
                        dtype: torch.dtype = torch.bfloat16) -> np.ndarray:
    """Host-side repack of a Conv3d [Co,Ci,3,3,3] / ConvTranspose3d [Ci,Co,3,3,3] weight into the MFMA
    fragment order of the conv kernel (uint16 bit patterns of ``dtype`` = bf16 or fp16)."""
    w = np.ascontiguousarray(weight.detach().to("cpu", torch.float32).numpy())
    if w.ndim != 5 or w.shape[2:] != (3, 3, 3):
        raise ValueError(f"pscv: conv3d weights must be [*,*,3,3,3], got {w.shape}")
    c_in, c_out = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
    lib = L.lib()
    code = _TORCH2PSCV.get(dtype, -1)
    n = lib.pscv_pack_conv3d_weights(None, c_in, c_out, kind, int(transposed), code, None)
    if n < 0:
        L.check(int(n), "pscv_pack_conv3d_weights")
    packed = np.empty(n, dtype=np.uint16)
    n2 = lib.pscv_pack_conv3d_weights(w.ctypes.data_as(C.c_void_p), c_in, c_out, kind, int(transposed), code,
                                      packed.ctypes.data_as(C.c_void_p))
    if n2 != n:
        L.check(-1 if n2 >= 0 else int(n2), "pscv_pack_conv3d_weights")
    return packed


def pack_conv3d_weights_device(weight: torch.Tensor, kind: int, transposed: bool, dtype: torch.dtype) -> torch.Tensor:
    """Same repack as ``pack_conv3d_weights`` in one launch on the GPU (pscv_pack_conv3d_weights_device): weights that
    change every optimizer step never travel to the host.  Returns the packed int16 tensor on the weight's device."""
    w = weight.detach().to(torch.float32).contiguous()
    _dev(w)
    if w.dim() != 5 or tuple(w.shape[2:]) != (3, 3, 3):
        raise ValueError(f"pscv: conv3d weights must be [*,*,3,3,3], got {tuple(w.shape)}")
    c_in, c_out = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
    code = _TORCH2PSCV.get(dtype, -1)
    n = L.lib().pscv_pack_conv3d_weights(None, c_in, c_out, kind, int(transposed), code, None)
    if n < 0:
        L.check(int(n), "pscv_pack_conv3d_weights")
    packed = torch.empty((n,), dtype=torch.int16, device=w.device)
    rc = _launch("pack_conv3d_weights", lambda: L.lib().pscv_pack_conv3d_weights_device(_p(w), c_in, c_out, kind, int(transposed), code,
                                                                                       _p(packed), _stream()))
    L.check(rc, "pscv_pack_conv3d_weights_device")
    return packed


@dataclass
class Conv3dLayer:
    """One 3x3x3 layer ready for the engine: packed 16-bit weights + fp32 epilogue vectors, on device."""
    packed: torch.Tensor            # int16 bit patterns of the MFMA A fragments
    dtype: torch.dtype              # bf16 or fp16: format of the weights AND of the activations it reads
    c_in: int
    c_out: int
    kind: int
    epi: int
    scale: Optional[torch.Tensor] = None
    bias: Optional[torch.Tensor] = None
    floor: Optional[torch.Tensor] = None

    @staticmethod
    def build(weight: torch.Tensor, *, kind: int, transposed: bool = False, device=None,
              bn: Optional[Sequence[torch.Tensor]] = None, bn_eps: float = 1e-5,
              conv_bias: Optional[torch.Tensor] = None, relu: bool = False, relu_post: bool = False,
              floor: Optional[torch.Tensor] = None, dtype: torch.dtype = torch.bfloat16) -> "Conv3dLayer":
        """``bn`` = (gamma, beta, running_mean, running_var) folds an eval-mode BatchNorm3d into the
        epilogue: scale = gamma / sqrt(var + eps), bias = beta - mean * scale (+ scale * conv_bias)."""
        device = device if device is not None else weight.device
        c_in, c_out = (weight.shape[0], weight.shape[1]) if transposed else (weight.shape[1], weight.shape[0])
        if kind == L.CONV_S1 and ((c_in in (8, 16, 32) and c_out == 8) or (SWEEP16 and (c_in, c_out) == (16, 16))) and USE_SWEEP_KERNEL:
            kind = L.CONV_S1P8     # same result, depth-sweep kernels with plane-pair packed MFMA rows (stride-1 deconvs too)
        # (16 -> 16 has a sweep variant too (kind S1P8 through the C ABI; ``SWEEP16 = True`` selects it), but since the brick kernel's staging / decode rewrite
        #  the brick kernel is faster at every measured size: 295 vs 337 us at 8x1024x1280, 16.6 vs 18.7 us at 96x64x80)
        if kind == L.CONV_T2 and transposed and c_in == 16 and c_out == 8 and USE_SWEEP_KERNEL:
            kind = L.CONV_T2P8     # same result, parity-pair packed MFMA rows + contiguous 32-byte stores
        if kind == L.CONV_S1 and not transposed and c_in in (8, 16) and c_out == 1 and USE_SWEEP_KERNEL:
            kind = L.CONV_S1C1     # same result, depth-in-rows MFMA kernel for the 1-channel heads
        # A training step applies the same Parameter many times (Vis-MVSNet: one pair U-Net per source view and stage, forward and
        # adjoint): raw-weight layers of live Parameters are memoised per parameter VERSION (the optimiser's in-place update bumps
        # it), guarded by a weak reference so that a recycled id() can never alias another tensor.
        ckey = None
        if PACK_CACHE and isinstance(weight, torch.nn.Parameter) and weight.is_cuda and bn is None and conv_bias is None and floor is None:
            ckey = (_weights_epoch, id(weight), kind, bool(transposed), dtype, str(device), bool(relu), bool(relu_post))
            ent = _layer_cache.get(ckey)
            if ent is not None and ent[0]() is weight and ent[1] == weight._version:
                return ent[2]
        if weight.is_cuda and torch.device(device).type == "cuda":
            packed = pack_conv3d_weights_device(weight, kind, transposed, dtype)
        else:
            packed = torch.from_numpy(pack_conv3d_weights(weight, kind, transposed, dtype).view(np.int16)).to(device)
        scale = bias = None
        if bn is not None:
            gamma, beta, mean, var = [t.detach().to(device, torch.float32) for t in bn]
            scale = gamma / torch.sqrt(var + bn_eps)
            bias = beta - mean * scale
            if conv_bias is not None:
                bias = bias + scale * conv_bias.detach().to(device, torch.float32)
            scale, bias = scale.contiguous(), bias.contiguous()
        elif conv_bias is not None:
            bias = conv_bias.detach().to(device, torch.float32).contiguous()
        epi = (L.EPI_RELU_PRE if relu else 0) | (L.EPI_RELU_POST if relu_post else 0)
        if floor is not None:
            floor = floor.detach().to(device, torch.float32).contiguous()
        layer = Conv3dLayer(packed, dtype, int(c_in), int(c_out), kind, epi, scale, bias, floor)
        if ckey is not None:
            if len(_layer_cache) > 4096:
                _layer_cache.clear()
            _layer_cache[ckey] = (weakref.ref(weight), weight._version, layer)
        return layer


USE_SWEEP_KERNEL = True   # tests flip this to compare the two stride-1 kernels
SWEEP16 = False           # 16 -> 16 layers on the depth-sweep kernel (tests / measurements; the brick kernel is the faster one)
_weights_epoch = 0


def weights_epoch() -> int:
    """Generation counter folded into every packed-weight / folded-BatchNorm cache key of the package."""
    return _weights_epoch


def invalidate_weight_caches() -> None:
    """Drop every packed-weight, folded-BatchNorm and soft-min temperature cache of the process.

    The caches are keyed on (storage address, ``tensor._version``): in-place tensor ops, optimiser steps and
    ``load_state_dict`` bump the version and are picked up automatically.  Writes THROUGH ``.data`` (``p.data.copy_(ema)``,
    ``p.data.clamp_()``, old-style optimisers) do not bump it -- call this function (also exported as
    ``wild_deep_mvs_amd.invalidate()``) after such a write, or the engine keeps running the previously packed weights."""
    global _weights_epoch
    _weights_epoch += 1
    _layer_cache.clear()

PACK_CACHE = True         # memoise raw-weight layers of live Parameters per parameter version (Conv3dLayer.build)
_layer_cache: dict = {}


def conv_out_shape(kind: int, D: int, H: int, W: int):
    if kind in (L.CONV_S1, L.CONV_S1P8, L.CONV_S1C1):
        return D, H, W
    if kind == L.CONV_S2:
        return (D + 1) // 2, (H + 1) // 2, (W + 1) // 2
    return 2 * D, 2 * H, 2 * W   # CONV_T2, CONV_T2P8


def _conv_cost(vox_in: int, vox_out: int, c_in: int, c_out: int, taps: int, transposed_s2: bool, out_bytes: int, with_skip: bool):
    """(algorithmic HBM bytes, flops) of one convolution launch: input and output once (+ the skip tensor), 2 x taps x c_in x c_out
    per output voxel (a stride-2 transposed layer touches every input voxel with every tap instead)."""
    b = vox_in * c_in * 2 + vox_out * c_out * out_bytes + (vox_out * c_out * 2 if with_skip else 0)
    f = 2.0 * taps * c_in * c_out * (vox_in if transposed_s2 else vox_out)
    return float(b), f


def conv3d(x: torch.Tensor, layer: Conv3dLayer, *, skip: Optional[torch.Tensor] = None, in_coff: int = 0,
           skip_coff: int = 0, out: Optional[torch.Tensor] = None, out_coff: int = 0,
           out_dtype: Optional[torch.dtype] = None, x2: Optional[torch.Tensor] = None, x2_coff: int = 0) -> torch.Tensor:
    """x [B,D,H,W,Cs] in the layer's 16-bit format (reads channels [in_coff, in_coff+c_in)) ->
    [B,Do,Ho,Wo,c_out] in the same format or fp32 (or writes the channel slice [out_coff, out_coff+c_out)
    of ``out``).  With ``x2`` (16-channel depth-sweep layers) the input is the channel concatenation of
    x[..., in_coff:in_coff+8] and x2[..., x2_coff:x2_coff+8], gathered while staging (pscv_conv3d_cat2)."""
    _dev(x, skip, out, layer.packed)
    if x.dtype != layer.dtype or x.dim() != 5:
        raise TypeError(f"pscv.conv3d: input must be a {layer.dtype} [B,D,H,W,C] volume, got {x.dtype} {tuple(x.shape)}")
    out_dtype = layer.dtype if out_dtype is None else out_dtype
    B, D, H, W, cs = x.shape
    Do, Ho, Wo = conv_out_shape(layer.kind, D, H, W)
    if out is None:
        out = torch.empty((B, Do, Ho, Wo, layer.c_out), dtype=out_dtype, device=x.device)
    if tuple(out.shape[:4]) != (B, Do, Ho, Wo):
        raise ValueError(f"pscv.conv3d: out has shape {tuple(out.shape)}, expected [B,{Do},{Ho},{Wo},*]")
    if skip is not None and (skip.dtype != layer.dtype or tuple(skip.shape[:4]) != (B, Do, Ho, Wo)):
        raise ValueError("pscv.conv3d: skip must have the layer's dtype and the output's spatial shape")
    if x2 is not None:
        # the 16 input channels are cat([x[..., in_coff:in_coff+8], x2[..., x2_coff:x2_coff+8]]) -- never materialised
        _dev(x2)
        if layer.kind != L.CONV_S1P8 or layer.c_in != 16 or x2.dtype != layer.dtype or tuple(x2.shape[:4]) != (B, D, H, W):
            raise ValueError("pscv.conv3d: a second input needs a depth-sweep (S1P8) layer with c_in = 16 and a volume of x's "
                             f"spatial shape in the layer's dtype (layer kind {layer.kind}, c_in {layer.c_in}, x2 {tuple(x2.shape)})")
        rc = _launch(f"conv3d[8+8->{layer.c_out},k{layer.kind}]", lambda: L.lib().pscv_conv3d_cat2(
            _p(x), cs, in_coff, _p(x2), x2.shape[4], x2_coff, _dt(x), _p(layer.packed), _p(layer.scale), _p(layer.bias),
            _p(layer.floor), _p(skip), 0 if skip is None else skip.shape[4], skip_coff, _p(out), out.shape[4], out_coff, _dt(out),
            B, D, H, W, layer.c_out, layer.epi, _stream()),
            cost=lambda: _conv_cost(B * D * H * W, B * Do * Ho * Wo, 16, layer.c_out, 27, False, out.element_size(), skip is not None))
        L.check(rc, "pscv_conv3d_cat2")
        return out
    rc = _launch(f"conv3d[{layer.c_in}->{layer.c_out},k{layer.kind}]", lambda: L.lib().pscv_conv3d(
        _p(x), _dt(x), cs, in_coff, _p(layer.packed), _p(layer.scale), _p(layer.bias), _p(layer.floor), _p(skip),
        0 if skip is None else skip.shape[4], skip_coff, _p(out), out.shape[4], out_coff, _dt(out), B, D, H, W,
        layer.c_in, layer.c_out, layer.kind, layer.epi, _stream()),
        cost=lambda: _conv_cost(B * D * H * W, B * Do * Ho * Wo, layer.c_in, layer.c_out, 27, layer.kind in (L.CONV_T2, getattr(L, "CONV_T2P8", -1)),
                                out.element_size(), skip is not None))
    L.check(rc, "pscv_conv3d")
    return out


def conv3d_block8(x: torch.Tensor, layer1: Conv3dLayer, layer2: Conv3dLayer, *, residual: bool = True, in_coff: int = 0,
                  out: Optional[torch.Tensor] = None, out_coff: int = 0) -> Optional[torch.Tensor]:
    """Two stacked 8 -> 8 depth-sweep layers (+ the block input as residual) in one launch (pscv_conv3d_block8): the values of
    ``conv3d(conv3d(x, layer1), layer2, skip=x)`` without the intermediate volume.  Returns None when the layers are not both
    8 -> 8 depth-sweep (S1P8) layers in x's dtype (run the two launches then)."""
    _dev(x, out, layer1.packed, layer2.packed)
    if not (layer1.kind == layer2.kind == L.CONV_S1P8 and layer1.c_in == layer1.c_out == layer2.c_in == layer2.c_out == 8
            and layer1.dtype == layer2.dtype == x.dtype and x.dim() == 5):
        return None
    B, D, H, W, cs = x.shape
    if out is None:
        out = torch.empty((B, D, H, W, 8), dtype=x.dtype, device=x.device)
    if tuple(out.shape[:4]) != (B, D, H, W) or out.dtype != x.dtype:
        raise ValueError(f"pscv.conv3d_block8: out has shape {tuple(out.shape)} / {out.dtype}, expected [B,{D},{H},{W},*] {x.dtype}")
    vox = B * D * H * W
    rc = _launch("conv3d_block8", lambda: L.lib().pscv_conv3d_block8(
        _p(x), _dt(x), cs, in_coff, _p(layer1.packed), _p(layer1.scale), _p(layer1.bias), _p(layer1.floor), layer1.epi,
        _p(layer2.packed), _p(layer2.scale), _p(layer2.bias), _p(layer2.floor), layer2.epi, int(residual),
        _p(out), out.shape[4], out_coff, B, D, H, W, _stream()),
        cost=lambda: (float(vox * 32), 2.0 * 2 * 27 * 8 * 8 * vox))
    L.check(rc, "pscv_conv3d_block8")
    return out


# --------------------------------------------------------------------------------------------
# Vis-MVSNet UncertNet (eval mode), one launch
# --------------------------------------------------------------------------------------------
def pack_uncert_params(w1, bn1, w2, bn2, head, eps1: float = 1e-5, eps2: float = 1e-5) -> torch.Tensor:
    """The 752-float parameter block of ``pscv_uncert_net`` (include/pscv.h) from the module's tensors: conv weights
    [8,1,3,3], [8,8,3,3], [1,8,3,3]; ``bn*`` = (gamma, beta, running_mean, running_var), folded in fp32."""
    f = lambda t: t.detach().to(torch.float32)

    def fold(bn, eps):
        g, b, m, v = [f(t) for t in bn]
        s = g / torch.sqrt(v + eps)
        return s, b - m * s
    s1, b1 = fold(bn1, eps1)
    s2, b2 = fold(bn2, eps2)
    return torch.cat([f(w1).reshape(8, 9).t().reshape(-1), s1, b1,                     # [tap][co]
                      f(w2).reshape(8, 8, 9).permute(1, 2, 0).reshape(-1), s2, b2,       # [ci][tap][co]
                      f(head).reshape(8, 9).reshape(-1)]).contiguous()                   # [ci][tap]


def uncert_net(entropy: torch.Tensor, params: torch.Tensor) -> torch.Tensor:
    """entropy [N,H,W] fp32 -> log-uncertainty [N,H,W] fp32 (reference model_cas.py:77-98, eval mode)."""
    _dev(entropy, params)
    if entropy.dtype != torch.float32 or entropy.dim() != 3 or not entropy.is_contiguous():
        raise ValueError("pscv.uncert_net: entropy must be a contiguous fp32 [N,H,W] map")
    if params.dtype != torch.float32 or params.numel() != 752 or not params.is_contiguous():
        raise ValueError("pscv.uncert_net: params must be the 752-float block of pack_uncert_params")
    N, H, W = entropy.shape
    out = torch.empty_like(entropy)
    rc = _launch("uncert_net", lambda: L.lib().pscv_uncert_net(_p(entropy), _p(params), _p(out), N, H, W, _stream()))
    L.check(rc, "pscv_uncert_net")
    return out


# --------------------------------------------------------------------------------------------
# softargmin
# --------------------------------------------------------------------------------------------
def softargmin(logits: torch.Tensor, depth: Optional[torch.Tensor] = None, *, want_index: bool = False,
               want_conf: bool = False, conf_mode: int = 0, window: float = 2.0, want_entropy: bool = False,
               want_prob: bool = False, want_partials: bool = False, index_offset: int = 0, into: Optional[dict] = None) -> dict:
    """logits [B,D,h,w] (fp32 or bf16); depth [B,D] or [B,D,h,w] fp32.  Returns a dict with the requested
    maps, each [B,h,w] fp32 (``prob`` [B,D,h,w], ``partials`` [B,4,h,w]).  ``into`` = {name: tensor}: write those maps into
    the caller's contiguous fp32 tensors (e.g. batch slices of one buffer for all source views) instead of new ones."""
    _dev(logits, depth)
    if logits.dim() != 4:
        raise ValueError("pscv.softargmin: logits must be [B,D,h,w]")
    B, D, h, w = logits.shape
    per_pixel = depth is not None and depth.dim() == 4
    if depth is not None:
        if depth.dtype != torch.float32 or depth.shape[:2] != (B, D):
            raise ValueError("pscv.softargmin: depth must be fp32 [B,D] or [B,D,h,w]")
    mk = lambda *s: torch.empty(s, dtype=torch.float32, device=logits.device)
    o = {
        "depth": mk(B, h, w) if depth is not None else None,
        "index": mk(B, h, w) if want_index else None,
        "conf": mk(B, h, w) if want_conf else None,
        "entropy": mk(B, h, w) if want_entropy else None,
        "prob": mk(B, D, h, w) if want_prob else None,
        "partials": mk(B, 4, h, w) if want_partials else None,
    }
    for k, tgt in (into or {}).items():
        if o.get(k) is None or tgt.dtype != torch.float32 or tgt.shape != o[k].shape or not tgt.is_contiguous() or tgt.device != logits.device:
            raise ValueError(f"pscv.softargmin: into[{k!r}] must be a requested map's contiguous fp32 tensor of shape {tuple(o[k].shape) if o.get(k) is not None else None}")
        o[k] = tgt
    rc = _launch("softargmin", lambda: L.lib().pscv_softargmin(
        _p(logits), _dt(logits), _p(depth), 0 if depth is None else depth.stride(0), int(per_pixel), _p(o["depth"]),
        _p(o["index"]), _p(o["conf"]), _p(o["entropy"]), _p(o["prob"]), _p(o["partials"]), conf_mode, float(window),
        index_offset, B, D, h, w, _stream()))
    L.check(rc, "pscv_softargmin")
    return {k: v for k, v in o.items() if v is not None}


_tail_ws: dict = {}       # (device, stream) -> fp32 scratch: launches on different streams must not share partial-sum buffers


def _tail_workspace(device, n: int) -> torch.Tensor:
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _tail_ws.get(key)
    if ws is None or ws.numel() < n:
        ws = torch.empty(n, dtype=torch.float32, device=device)
        _tail_ws[key] = ws
    return ws


def prob_softargmin(x: torch.Tensor, layer: "Conv3dLayer", depth: torch.Tensor, *, in_coff: int = 0, want_conf: bool = True):
    """Fused tail of the MVSNet regulariser (pscv_prob_softargmin): x [B,D,h,w,Cs] 16-bit -> the 1-channel head ``layer`` (kind
    S1C1, c_in 8) -> fp32 logits [B,D,h,w] + depth [B,h,w] (+ 4-plane confidence), with per-batch depth planes ``depth`` [B,D].
    Returns None when the layer / size does not run the depth-sweep head (call ``conv3d`` + ``softargmin`` then)."""
    _dev(x, depth, layer.packed)
    if layer.kind != L.CONV_S1C1 or layer.c_in != 8 or x.dtype != layer.dtype or x.dim() != 5 or depth.dim() != 2 \
            or depth.dtype != torch.float32:
        return None
    B, D, H, W, cs = x.shape
    if tuple(depth.shape) != (B, D) or D < 18:
        return None
    n = int(L.lib().pscv_prob_softargmin_workspace(B, D, H, W))
    ws = _tail_workspace(x.device, n)
    logits = torch.empty((B, D, H, W), dtype=torch.float32, device=x.device)
    o_depth = torch.empty((B, H, W), dtype=torch.float32, device=x.device)
    o_conf = torch.empty((B, H, W), dtype=torch.float32, device=x.device) if want_conf else None
    rc = _launch("prob_softargmin", lambda: L.lib().pscv_prob_softargmin(
        _p(x), _dt(x), cs, in_coff, _p(layer.packed), _p(layer.scale), _p(layer.bias), _p(layer.floor), layer.c_in, layer.epi,
        _p(depth), depth.stride(0), _p(logits), _p(ws), ws.numel(), _p(o_depth), _p(o_conf), B, D, H, W, _stream()))
    if rc == -3:
        return None
    L.check(rc, "pscv_prob_softargmin")
    out = {"logits": logits, "depth": o_depth}
    if want_conf:
        out["conf"] = o_conf
    return out


def tail_sweep(x: torch.Tensor, up: "Conv3dLayer", head: "Conv3dLayer", *, skip: Optional[torch.Tensor] = None, in_coff: int = 0,
               skip_coff: int = 0, regress: Optional[torch.Tensor] = None, want_conf: bool = True):
    """Fused tail of the MVSNet regulariser (pscv_tail_sweep): x [B,Di,Hi,Wi,Cs] 16-bit -> transposed layer ``up`` (kind T2P8,
    16 -> 8, + ``skip`` [B,2Di,2Hi,2Wi,*]) -> 1-channel head ``head`` (kind S1C1) -> fp32 logits [B,2Di,2Hi,2Wi]; the 8-channel
    full-resolution volume is never stored.  Same bits as ``conv3d(conv3d(x, up, skip=skip), head, out_dtype=float32)``.
    With ``regress`` = per-batch depth planes [B,2Di] the sweep also keeps softmax statistics and a merge launch regresses depth
    (and the 4-plane confidence): returns {"logits", "depth", "conf"} -- no separate softargmin pass.  Returns None when the
    layers / shape are not covered (run the two layers then)."""
    _dev(x, skip, up.packed, head.packed, regress)
    if (up.kind != L.CONV_T2P8 or head.kind != L.CONV_S1C1 or up.c_in != 16 or up.c_out != 8 or head.c_in != 8 or x.dim() != 5
            or x.dtype != up.dtype or head.dtype != up.dtype or (skip is not None and skip.dtype != x.dtype)):
        return None
    B, Di, Hi, Wi, cs = x.shape
    if skip is not None and tuple(skip.shape[:4]) != (B, 2 * Di, 2 * Hi, 2 * Wi):
        raise ValueError(f"pscv.tail_sweep: skip has shape {tuple(skip.shape)}, expected [B,{2 * Di},{2 * Hi},{2 * Wi},*]")
    if regress is not None and (regress.dtype != torch.float32 or tuple(regress.shape) != (B, 2 * Di)):
        raise ValueError(f"pscv.tail_sweep: regress must be fp32 depth planes [B,{2 * Di}]")
    logits = torch.empty((B, 2 * Di, 2 * Hi, 2 * Wi), dtype=torch.float32, device=x.device)
    ws = o_depth = o_conf = None
    if regress is not None:
        ws = _tail_workspace(x.device, int(L.lib().pscv_tail_sweep_workspace(B, Di, Hi, Wi)))
        o_depth = torch.empty((B, 2 * Hi, 2 * Wi), dtype=torch.float32, device=x.device)
        o_conf = torch.empty((B, 2 * Hi, 2 * Wi), dtype=torch.float32, device=x.device) if want_conf else None
    vox = B * 8 * Di * Hi * Wi
    rc = _launch("tail_sweep", lambda: L.lib().pscv_tail_sweep(
        _p(x), _dt(x), cs, in_coff, _p(up.packed), _p(up.scale), _p(up.bias), _p(up.floor), up.epi, _p(skip),
        skip.shape[4] if skip is not None else 0, skip_coff, _p(head.packed), _p(head.scale), _p(head.bias), _p(head.floor), head.epi,
        _p(logits), _p(regress), regress.stride(0) if regress is not None else 0, _p(ws), ws.numel() if ws is not None else 0,
        _p(o_depth), _p(o_conf), B, Di, Hi, Wi, _stream()),
        cost=lambda: (vox // 8 * 32 + vox * (16 if skip is not None else 0) + vox * 4, 2.0 * vox * 27 * 8 + 2.0 * (vox // 8) * 27 * 16 * 8))
    if rc == 1:
        return None
    L.check(rc, "pscv_tail_sweep")
    if regress is None:
        return logits
    out = {"logits": logits, "depth": o_depth}
    if want_conf:
        out["conf"] = o_conf
    return out


def head_index_entropy(x: torch.Tensor, layer: "Conv3dLayer", index: torch.Tensor, entropy: torch.Tensor, *, want_scores: bool = False):
    """Fused head of a Vis pair branch (pscv_head_index_entropy): x [B,D,h,w,8] 16-bit -> the 1-channel head ``layer`` (kind S1C1)
    -> expected plane index and entropy written into the caller's fp32 [B,h,w] tensors; the fp32 scores only with
    ``want_scores``.  Returns None when the layer / size does not run the depth-sweep head (call ``conv3d`` + ``softargmin``
    then), else the scores or True."""
    _dev(x, layer.packed, index, entropy)
    if layer.kind != L.CONV_S1C1 or layer.c_in != 8 or x.dtype != layer.dtype or x.dim() != 5:
        return None
    B, D, H, W, cs = x.shape
    for t_ in (index, entropy):
        if t_.dtype != torch.float32 or tuple(t_.shape) != (B, H, W) or not t_.is_contiguous():
            raise ValueError("pscv.head_index_entropy: index / entropy must be contiguous fp32 [B,h,w] tensors")
    n = int(L.lib().pscv_prob_softargmin_workspace(B, D, H, W))
    ws = _tail_workspace(x.device, n)
    scores = torch.empty((B, D, H, W), dtype=torch.float32, device=x.device) if want_scores else None
    rc = _launch("head_index_entropy", lambda: L.lib().pscv_head_index_entropy(
        _p(x), _dt(x), cs, 0, _p(layer.packed), _p(layer.scale), _p(layer.bias), _p(layer.floor), layer.c_in, layer.epi,
        _p(scores), _p(ws), ws.numel(), _p(index), _p(entropy), B, D, H, W, _stream()),
        cost=lambda: (B * D * H * W * (16 + (4 if want_scores else 0)), 2.0 * B * D * H * W * 27 * 8))
    if rc == -3:
        return None
    L.check(rc, "pscv_head_index_entropy")
    return scores if want_scores else True


def softargmin_window(logits: torch.Tensor, stats: torch.Tensor, *, window: float, index_offset: int) -> torch.Tensor:
    """One depth shard's part of the +-window probability under globally merged softmax statistics: logits fp32 [B,D,h,w] (this
    shard's planes), stats fp32 [B,3,h,w] = (max, sum exp, expected index) -> fp32 [B,h,w] (pscv_softargmin_window)."""
    _dev(logits, stats)
    B, D, h, w = logits.shape
    if logits.dtype != torch.float32 or stats.dtype != torch.float32 or tuple(stats.shape) != (B, 3, h, w):
        raise ValueError("pscv.softargmin_window: fp32 logits [B,D,h,w] and stats [B,3,h,w] expected")
    out = torch.empty((B, h, w), dtype=torch.float32, device=logits.device)
    rc = _launch("softargmin_window", lambda: L.lib().pscv_softargmin_window(_p(logits), _p(stats), _p(out), float(window),
                                                                            int(index_offset), B, D, h, w, _stream()))
    L.check(rc, "pscv_softargmin_window")
    return out


# --------------------------------------------------------------------------------------------
# unsupervised photometric loss: depth -> flows -> warped sources, SSIM
# --------------------------------------------------------------------------------------------
def inv_proj4x4(P: torch.Tensor) -> torch.Tensor:
    """Inverse of projection matrices [...,4,4] with last row (0,0,0,1): [[A^-1, -A^-1 b], [0 0 0 1]], closed form in fp64
    (the reference calls torch.inverse in fp32, utils_3D.py:196; no LAPACK call / host sync here)."""
    Pd = P.double()
    Ai = inv3x3(Pd[..., :3, :3])
    out = torch.zeros_like(Pd)
    out[..., :3, :3] = Ai
    out[..., :3, 3:] = -(Ai @ Pd[..., :3, 3:])
    out[..., 3, 3] = 1.0
    return out.to(P.dtype)


def _photo_args(src_imgs, depth, inv_ref, proj_src):
    _dev(depth, inv_ref, proj_src)
    B, h, w = depth.shape
    S = proj_src.shape[1]
    if depth.dtype != torch.float32 or inv_ref.dtype != torch.float32 or proj_src.dtype != torch.float32 \
            or tuple(inv_ref.shape) != (B, 4, 4) or tuple(proj_src.shape) != (B, S, 4, 4):
        raise ValueError("pscv.photo_warp: fp32 depth [B,h,w], inv_ref [B,4,4], proj_src [B,S,4,4] expected")
    C = 0
    if src_imgs is not None:
        _dev(src_imgs)
        C = src_imgs.shape[2]
        if src_imgs.dtype != torch.float32 or tuple(src_imgs.shape) != (B, S, C, h, w):
            raise ValueError(f"pscv.photo_warp: source images fp32 [B,S,C,h,w] at the depth map's resolution expected, got "
                             f"{tuple(src_imgs.shape)} for depth {tuple(depth.shape)}")
    return B, S, C, h, w


def photo_warp(src_imgs: Optional[torch.Tensor], depth: torch.Tensor, inv_ref: torch.Tensor, proj_src: torch.Tensor, *,
               src_depth: Optional[torch.Tensor] = None, want_mask: bool = True, want_z: bool = False, want_flows: bool = False) -> dict:
    """Depth map -> flows -> warped sources (pscv_photo_warp; trainer.py:209-236).  Returns a dict with ``warped`` [B,S,C,h,w]
    (if images are given), ``mask`` [B,S,h,w] fp32, ``z`` [B,S,h,w], ``flows`` [B,S,h,w,2], ``warped_depth`` [B,S,h,w]."""
    B, S, C, h, w = _photo_args(src_imgs, depth, inv_ref, proj_src)
    depth, inv_ref, proj_src = depth.contiguous(), inv_ref.contiguous(), proj_src.contiguous()
    mk = lambda *sh: torch.empty(sh, dtype=torch.float32, device=depth.device)
    o = {"warped": mk(B, S, C, h, w) if src_imgs is not None else None, "mask": mk(B, S, h, w) if want_mask else None,
         "z": mk(B, S, h, w) if want_z else None, "flows": mk(B, S, h, w, 2) if want_flows else None,
         "warped_depth": mk(B, S, h, w) if src_depth is not None else None}
    if src_imgs is not None:
        src_imgs = src_imgs.contiguous()
    if src_depth is not None:
        _dev(src_depth)
        if src_depth.dtype != torch.float32 or tuple(src_depth.shape) != (B, S, h, w):
            raise ValueError("pscv.photo_warp: src_depth fp32 [B,S,h,w] expected")
        src_depth = src_depth.contiguous()
    rc = _launch("photo_warp", lambda: L.lib().pscv_photo_warp(_p(src_imgs), _p(depth), _p(inv_ref), _p(proj_src), _p(src_depth),
                                                              _p(o["warped"]), _p(o["mask"]), _p(o["z"]), _p(o["flows"]),
                                                              _p(o["warped_depth"]), B, S, C, h, w, _stream()))
    L.check(rc, "pscv_photo_warp")
    return o


def photo_warp_bwd(src_imgs: torch.Tensor, depth: torch.Tensor, inv_ref: torch.Tensor, proj_src: torch.Tensor,
                   grad_warped: torch.Tensor) -> torch.Tensor:
    """grad_warped [B,S,C,h,w] -> grad_depth [B,h,w] (pscv_photo_warp_bwd)."""
    B, S, C, h, w = _photo_args(src_imgs, depth, inv_ref, proj_src)
    _dev(grad_warped)
    if grad_warped.dtype != torch.float32 or tuple(grad_warped.shape) != (B, S, C, h, w):
        raise ValueError("pscv.photo_warp_bwd: grad_warped fp32 [B,S,C,h,w] expected")
    gd = torch.empty((B, h, w), dtype=torch.float32, device=depth.device)
    src_imgs, depth, inv_ref, proj_src, grad_warped = (t.contiguous() for t in (src_imgs, depth, inv_ref, proj_src, grad_warped))
    rc = _launch("photo_warp_bwd", lambda: L.lib().pscv_photo_warp_bwd(_p(src_imgs), _p(depth), _p(inv_ref), _p(proj_src),
                                                                      _p(grad_warped), _p(gd), B, S, C, h, w, _stream()))
    L.check(rc, "pscv_photo_warp_bwd")
    return gd


def _ssim_args(img1, img2):
    _dev(img1, img2)
    if img1.dtype != torch.float32 or img2.dtype != torch.float32 or img1.dim() != 4 or img2.dim() != 4 \
            or img1.shape[1:] != img2.shape[1:] or img2.shape[0] % img1.shape[0]:
        raise ValueError(f"pscv.ssim: fp32 img1 [n1,C,h,w] and img2 [n1*rep,C,h,w] expected, got {tuple(img1.shape)} {tuple(img2.shape)}")
    n1, C, h, w = img1.shape
    return n1, img2.shape[0] // n1, C, h, w


def ssim(img1: torch.Tensor, img2: torch.Tensor) -> torch.Tensor:
    """1 - SSIM per channel (pscv_ssim; utils/ssimLoss.py:27-60): img1 [n1,C,h,w], img2 [n1*rep,C,h,w] -> [n1*rep,C,h,w]."""
    n1, rep, C, h, w = _ssim_args(img1, img2)
    img1, img2 = img1.contiguous(), img2.contiguous()
    out = torch.empty_like(img2)
    rc = _launch("ssim", lambda: L.lib().pscv_ssim(_p(img1), _p(img2), _p(out), n1, rep, C, h, w, _stream()))
    L.check(rc, "pscv_ssim")
    return out


def ssim_bwd(img1: torch.Tensor, img2: torch.Tensor, grad_out: torch.Tensor) -> torch.Tensor:
    """Gradient of ``ssim`` to img2 (pscv_ssim_bwd)."""
    n1, rep, C, h, w = _ssim_args(img1, img2)
    _dev(grad_out)
    if grad_out.dtype != torch.float32 or grad_out.shape != img2.shape:
        raise ValueError("pscv.ssim_bwd: grad_out must match img2")
    img1, img2, grad_out = img1.contiguous(), img2.contiguous(), grad_out.contiguous()
    ws = torch.empty((3 * img2.numel(),), dtype=torch.float32, device=img2.device)
    g2 = torch.empty_like(img2)
    rc = _launch("ssim_bwd", lambda: L.lib().pscv_ssim_bwd(_p(img1), _p(img2), _p(grad_out), _p(ws), _p(g2), n1, rep, C, h, w, _stream()))
    L.check(rc, "pscv_ssim_bwd")
    return g2


# --------------------------------------------------------------------------------------------
# CVP refinement hypotheses
# --------------------------------------------------------------------------------------------
def cvp_depth_hypos(depth: torch.Tensor, cams: torch.Tensor, fallback: torch.Tensor, *, want_steps: bool = False):
    """depth fp32 [B,H,W], cams fp64 [B,39] (K_ref^-1, rows 0..2 of E_src E_ref^-1, K_src, (K_ref R_ref)(K_src R_src)^-1),
    fallback fp32 [B] -> hypotheses fp32 [B,8,H,W] = depth + k * median|step| (pscv_cvp_depth_hypos; no host sync)."""
    _dev(depth, cams, fallback)
    if depth.dtype != torch.float32 or depth.dim() != 3 or cams.dtype != torch.float64 or tuple(cams.shape) != (depth.shape[0], 39) \
            or fallback.dtype != torch.float32 or fallback.numel() != depth.shape[0]:
        raise ValueError("pscv.cvp_depth_hypos: depth fp32 [B,H,W], cams fp64 [B,39], fallback fp32 [B] expected")
    B, H, W = depth.shape
    keys = torch.empty((B * H * W + B * 1040,), dtype=torch.int64, device=depth.device)
    steps = torch.empty((B,), dtype=torch.float64, device=depth.device)
    hypos = torch.empty((B, 8, H, W), dtype=torch.float32, device=depth.device)
    rc = _launch("cvp_depth_hypos", lambda: L.lib().pscv_cvp_depth_hypos(_p(depth), _p(cams), _p(fallback), _p(keys), _p(steps),
                                                                        _p(hypos), B, H, W, _stream()))
    L.check(rc, "pscv_cvp_depth_hypos")
    return (hypos, steps) if want_steps else hypos


def homography_warp(image_cl: torch.Tensor, H: torch.Tensor, ref_hw: Sequence[int]) -> torch.Tensor:
    """image fp32 channels-last [m,hs,ws,c]; H fp32 [m,3,3] (one per batch item) or [m,h,w,3,3] (one per reference pixel)
    -> warped fp32 [m,h,w,c] (pscv_homography_warp)."""
    _dev(image_cl, H)
    if image_cl.dtype != torch.float32 or image_cl.dim() != 4 or H.dtype != torch.float32:
        raise TypeError("pscv.homography_warp: fp32 channels-last image [m,hs,ws,c] and fp32 homographies expected")
    m, hs, ws, c = image_cl.shape
    h, w = int(ref_hw[0]), int(ref_hw[1])
    per_pixel = H.dim() == 5
    if tuple(H.shape) not in ((m, 3, 3), (m, h, w, 3, 3)):
        raise ValueError(f"pscv.homography_warp: H must be [{m},3,3] or [{m},{h},{w},3,3], got {tuple(H.shape)}")
    out = torch.empty((m, h, w, c), dtype=torch.float32, device=image_cl.device)
    rc = _launch("homography_warp", lambda: L.lib().pscv_homography_warp(_p(image_cl), _p(H), int(per_pixel), _p(out), m, c, h, w, hs,
                                                                       ws, _stream()))
    L.check(rc, "pscv_homography_warp")
    return out


def homography_warp_bwd(grad_out: torch.Tensor, H: torch.Tensor, src_hw: Sequence[int]) -> torch.Tensor:
    """Adjoint of ``homography_warp`` w.r.t. the image: grad_out fp32 [m,h,w,c], H as in the forward -> grad_image fp32 [m,hs,ws,c]
    (pscv_homography_warp_bwd; reference: autograd of grid_sample's input, models/VisMVSNet/homography.py:101-102)."""
    _dev(grad_out, H)
    if grad_out.dtype != torch.float32 or grad_out.dim() != 4 or H.dtype != torch.float32:
        raise TypeError("pscv.homography_warp_bwd: fp32 channels-last gradient [m,h,w,c] and fp32 homographies expected")
    grad_out = grad_out.contiguous()
    m, h, w, c = grad_out.shape
    hs, ws = int(src_hw[0]), int(src_hw[1])
    per_pixel = H.dim() == 5
    if tuple(H.shape) not in ((m, 3, 3), (m, h, w, 3, 3)):
        raise ValueError(f"pscv.homography_warp_bwd: H must be [{m},3,3] or [{m},{h},{w},3,3], got {tuple(H.shape)}")
    gimg = torch.zeros((m, hs, ws, c), dtype=torch.float32, device=grad_out.device)
    rc = _launch("homography_warp_bwd", lambda: L.lib().pscv_homography_warp_bwd(_p(grad_out), _p(H), int(per_pixel), _p(gimg), m, c, h, w,
                                                                               hs, ws, _stream()))
    L.check(rc, "pscv_homography_warp_bwd")
    return gimg


def cvp_cams(ref_in: torch.Tensor, src_in: torch.Tensor, ref_ex: torch.Tensor, src_ex: torch.Tensor, level_scales: Sequence[float],
             *, want_hypo: bool = True):
    """All camera blocks of a CVP-MVSNet forward in one launch (pscv_cvp_cams): ref_in [B,3,3], src_in [B,N,3,3], ref_ex [B,4,4],
    src_ex [B,N,4,4]; ``level_scales[l]`` = image_height / level_height.  Returns (warp cams fp32 [L,N,B,18] -- ``warp[l]`` is the
    ``cams`` of ``warp_cost`` at level l --, hypothesis cams fp64 [L,B,39] for ``cvp_depth_hypos`` or None)."""
    f32 = lambda t: t.detach().to(torch.float32).contiguous()
    ref_in, src_in, ref_ex, src_ex = f32(ref_in), f32(src_in), f32(ref_ex), f32(src_ex)
    _dev(ref_in, src_in, ref_ex, src_ex)
    B, N = src_in.shape[:2]
    if tuple(ref_in.shape) != (B, 3, 3) or tuple(src_in.shape) != (B, N, 3, 3) or tuple(ref_ex.shape) != (B, 4, 4) \
            or tuple(src_ex.shape) != (B, N, 4, 4):
        raise ValueError("pscv.cvp_cams: ref_in [B,3,3], src_in [B,N,3,3], ref_ex [B,4,4], src_ex [B,N,4,4] expected")
    nl = len(level_scales)
    sc = (C.c_float * nl)(*[float(v) for v in level_scales])
    warp = torch.empty((nl, N, B, L.CAM_FLOATS), dtype=torch.float32, device=ref_in.device)
    hypo = torch.empty((nl, B, 39), dtype=torch.float64, device=ref_in.device) if want_hypo else None
    rc = _launch("cvp_cams", lambda: L.lib().pscv_cvp_cams(_p(ref_in), _p(src_in), _p(ref_ex), _p(src_ex), sc, B, N, nl, _p(warp),
                                                         _p(hypo), _stream()))
    L.check(rc, "pscv_cvp_cams")
    return warp, hypo


# --------------------------------------------------------------------------------------------
# training path: batch-statistics BatchNorm pieces, weight gradients, backward of the sweep
# --------------------------------------------------------------------------------------------
_train_ws = {}


def _workspace(device, nfloats: int) -> torch.Tensor:
    """fp32 scratch for the two-phase reductions, one per (device, stream), grown on demand: reuse is ordered by the stream, and
    launches on different streams (round-3 advisor finding: the batch stream mode) never share a partial-sum buffer."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _train_ws.get(key)
    if ws is None or ws.numel() < nfloats:
        ws = torch.empty(max(int(nfloats), int(L.lib().pscv_train_workspace_floats())), dtype=torch.float32, device=device)
        _train_ws[key] = ws
    return ws


def _vol16(x: torch.Tensor, what: str):
    if x.dtype not in HALF_DTYPES or x.dim() != 5 or x.shape[4] not in (8, 16, 32, 64, 128):
        raise TypeError(f"pscv.{what}: a 16-bit channels-last volume [B,D,h,w,C] with C in 8/16/32/64/128 is expected, got "
                        f"{x.dtype} {tuple(x.shape)}")


def _group_vox(y: torch.Tensor, groups: int, what: str) -> int:
    """Voxels per group of a contiguous volume whose leading (batch) axis holds ``groups`` equal consecutive slices."""
    if groups < 1 or y.shape[0] % groups or not y.is_contiguous():
        raise ValueError(f"pscv.{what}: {groups} groups need a contiguous volume whose batch axis ({y.shape[0]}) they divide")
    return y.numel() // y.shape[4] // groups


def _dev_rows(*ts: Optional[torch.Tensor]):
    """Per-channel constants: device tensors whose channel axis is dense (a [C] vector, or [G,C] rows of a larger tensor)."""
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("pscv: the plane-sweep engine runs on MI355X only; got a CPU tensor")
        if t.stride(-1) != 1 or (t.dim() == 1 and not t.is_contiguous()):
            raise ValueError("pscv: per-channel constants must be dense along the channel axis")


def _row_stride(a: torch.Tensor, b: torch.Tensor, groups: int, Cc: int, what: str, c: Optional[torch.Tensor] = None) -> int:
    """Per-group constants handed over as [G,C] views of one [G,k,C] tensor: the float stride between groups (the same for all)."""
    ts = [t for t in (a, b, c) if t is not None]
    for t in ts:
        if t.dtype != torch.float32 or t.shape != (groups, Cc) or t.stride(1) != 1 or t.stride(0) != ts[0].stride(0):
            raise ValueError(f"pscv.{what}: per-group constants must be fp32 [groups,C] rows with one common group stride")
    return int(ts[0].stride(0))


def bn_stats(y: torch.Tensor, groups: int = 1) -> torch.Tensor:
    """y [B,D,h,w,C] 16-bit -> fp32 [2,C]: per-channel sum and sum of squares over all voxels (pscv_bn_stats).  ``groups`` > 1:
    the batch axis is ``groups`` consecutive slices with their own statistics (the views of a 2-D extractor batch) -> [groups,2,C]."""
    _dev(y)
    _vol16(y, "bn_stats")
    Cc = y.shape[4]
    nv = _group_vox(y, groups, "bn_stats")
    sums = torch.empty((2, Cc) if groups == 1 else (groups, 2, Cc), dtype=torch.float32, device=y.device)
    ws = _workspace(y.device, 0)
    rc = _launch("bn_stats", lambda: L.lib().pscv_bn_stats_grouped(_p(y), _dt(y), nv, groups, Cc, _p(ws), _p(sums), _stream()))
    L.check(rc, "pscv_bn_stats")
    return sums


def bn_finalize(sums: torch.Tensor, nvox: int, bn) -> torch.Tensor:
    """sums fp32 [2,C] of ``bn_stats`` -> fp32 [4,C] = (scale, bias, mean, invstd) of a BatchNorm module in train(); updates its
    running statistics and ``num_batches_tracked`` like nn.BatchNorm3d (pscv_bn_finalize, one launch).  Grouped sums [G,2,C]
    (``nvox`` per group) -> [G,4,C]; the running statistics are updated once per group, in order, as G forward calls would."""
    _dev(sums)
    groups = 1 if sums.dim() == 2 else sums.shape[0]
    Cc = sums.shape[-1]
    out = torch.empty((4, Cc) if sums.dim() == 2 else (groups, 4, Cc), dtype=torch.float32, device=sums.device)
    track = bn.track_running_stats and bn.running_mean is not None
    if track and bn.momentum is None:
        raise NotImplementedError("pscv BatchNorm training: cumulative moving average (momentum=None) is not used by the reference")
    f32 = lambda t_: None if t_ is None else (t_ if t_.dtype == torch.float32 else t_.detach().float())
    gamma, beta = f32(bn.weight), f32(bn.bias)
    rm, rv = (bn.running_mean, bn.running_var) if track else (None, None)
    if track and (rm.dtype != torch.float32 or rv.dtype != torch.float32):
        raise TypeError("pscv BatchNorm training: fp32 running statistics expected")
    nbt = bn.num_batches_tracked if track else None
    rc = _launch("bn_finalize", lambda: L.lib().pscv_bn_finalize_grouped(_p(sums), int(nvox), groups, Cc, _p(gamma), _p(beta), float(bn.eps),
                                                                        float(bn.momentum if track else 0.0), _p(rm), _p(rv), _p(nbt), _p(out),
                                                                        _stream()))
    L.check(rc, "pscv_bn_finalize")
    return out


def bn_bwd_coeffs(sums: torch.Tensor, mean: torch.Tensor, invstd: torch.Tensor, gamma: Optional[torch.Tensor], nvox: int) -> torch.Tensor:
    """sums fp32 [2,C] of ``bn_bwd_reduce`` -> fp32 [5,C] = (ca, cb, cc, d gamma, d beta) (pscv_bn_bwd_coeffs, one launch).
    Grouped: sums [G,2,C], ``mean`` / ``invstd`` = rows 2 / 3 of ``bn_finalize``'s [G,4,C] (views) -> [G,5,C]."""
    _dev(sums, gamma)
    _dev_rows(mean, invstd)
    groups = 1 if sums.dim() == 2 else sums.shape[0]
    Cc = sums.shape[-1]
    out = torch.empty((5, Cc) if sums.dim() == 2 else (groups, 5, Cc), dtype=torch.float32, device=sums.device)
    g = None if gamma is None else (gamma.detach() if gamma.dtype == torch.float32 else gamma.detach().float())
    sst = 0 if groups == 1 else _row_stride(mean, invstd, groups, Cc, "bn_bwd_coeffs")
    rc = _launch("bn_bwd_coeffs", lambda: L.lib().pscv_bn_bwd_coeffs_grouped(_p(sums), _p(mean), _p(invstd), sst, _p(g), int(nvox), groups, Cc,
                                                                            _p(out), _stream()))
    L.check(rc, "pscv_bn_bwd_coeffs")
    return out


def bn_act(y: torch.Tensor, scale: torch.Tensor, bias: torch.Tensor, *, relu, skip: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[relu](y * scale + bias) + skip on a 16-bit channels-last volume (pscv_bn_act).  ``relu``: False / True (before the
    skip add) / "post" (after it: the Vis BasicBlock)."""
    _dev(y, skip)
    _dev_rows(scale, bias)
    _vol16(y, "bn_act")
    if skip is not None and (skip.shape != y.shape or skip.dtype != y.dtype):
        raise ValueError("pscv.bn_act: skip must match y")
    Cc = y.shape[4]
    groups = 1 if scale.dim() == 1 else scale.shape[0]          # grouped: scale / bias [G,C] (rows of bn_finalize's [G,4,C])
    nv = _group_vox(y, groups, "bn_act") if groups > 1 else y.numel() // Cc
    pst = 0 if groups == 1 else _row_stride(scale, bias, groups, Cc, "bn_act")
    out = torch.empty_like(y)
    rc = _launch("bn_act", lambda: L.lib().pscv_bn_act_grouped(_p(y), _dt(y), nv, groups, Cc, _p(scale), _p(bias), pst,
                                                              2 if relu == "post" else int(bool(relu)), _p(skip), _p(out), _stream()))
    L.check(rc, "pscv_bn_act")
    return out


def bn_bwd_reduce(dact: torch.Tensor, y: torch.Tensor, scale: torch.Tensor, bias: torch.Tensor, *, relu: bool) -> torch.Tensor:
    """fp32 [2,C]: sum dz and sum dz*y with dz = dact * [y*scale+bias > 0] (pscv_bn_bwd_reduce)."""
    _dev(dact, y)
    _dev_rows(scale, bias)
    _vol16(y, "bn_bwd_reduce")
    if dact.shape != y.shape or dact.dtype != y.dtype:
        raise ValueError("pscv.bn_bwd_reduce: dact must match y")
    Cc = y.shape[4]
    groups = 1 if scale.dim() == 1 else scale.shape[0]
    nv = _group_vox(y, groups, "bn_bwd_reduce") if groups > 1 else y.numel() // Cc
    pst = 0 if groups == 1 else _row_stride(scale, bias, groups, Cc, "bn_bwd_reduce")
    if not dact.is_contiguous():
        raise ValueError("pscv.bn_bwd_reduce: dact must be contiguous")
    sums = torch.empty((2, Cc) if scale.dim() == 1 else (groups, 2, Cc), dtype=torch.float32, device=y.device)
    ws = _workspace(y.device, 0)
    rc = _launch("bn_bwd_reduce", lambda: L.lib().pscv_bn_bwd_reduce_grouped(_p(dact), _p(y), _dt(y), nv, groups, Cc, _p(scale), _p(bias), pst,
                                                                            int(relu), _p(ws), _p(sums), _stream()))
    L.check(rc, "pscv_bn_bwd_reduce")
    return sums


def bn_bwd_apply(dact: torch.Tensor, y: torch.Tensor, scale: torch.Tensor, bias: torch.Tensor, ca: torch.Tensor, cb: torch.Tensor,
                 cc: torch.Tensor, *, relu: bool) -> torch.Tensor:
    """dy = ca * dz + cb * y + cc per channel (pscv_bn_bwd_apply)."""
    _dev(dact, y)
    _dev_rows(scale, bias, ca, cb, cc)
    _vol16(y, "bn_bwd_apply")
    Cc = y.shape[4]
    groups = 1 if scale.dim() == 1 else scale.shape[0]
    nv = _group_vox(y, groups, "bn_bwd_apply") if groups > 1 else y.numel() // Cc
    pst = 0 if groups == 1 else _row_stride(scale, bias, groups, Cc, "bn_bwd_apply")
    cst = 0 if groups == 1 else _row_stride(ca, cb, groups, Cc, "bn_bwd_apply", cc)
    dy = torch.empty_like(y)
    rc = _launch("bn_bwd_apply", lambda: L.lib().pscv_bn_bwd_apply_grouped(_p(dact), _p(y), _dt(y), nv, groups, Cc, _p(scale), _p(bias), pst,
                                                                          int(relu), _p(ca), _p(cb), _p(cc), cst, _p(dy), _stream()))
    L.check(rc, "pscv_bn_bwd_apply")
    return dy


def softargmin_bwd(logits: torch.Tensor, depth: Optional[torch.Tensor], grad_depth: Optional[torch.Tensor], dtype: torch.dtype, *,
                   grad_index: Optional[torch.Tensor] = None, grad_entropy: Optional[torch.Tensor] = None) -> torch.Tensor:
    """logits fp32 [B,D,h,w]; upstream gradients fp32 [B,h,w] of depth (needs the planes [B,D] / [B,D,h,w]), expected index
    and entropy (each optional) -> [B,D,h,w,8] in ``dtype`` with d loss / d logit in channel 0 and zeros elsewhere
    (pscv_softargmin_bwd)."""
    _dev(logits, depth, grad_depth, grad_index, grad_entropy)
    if logits.dtype != torch.float32 or logits.dim() != 4:
        raise TypeError("pscv.softargmin_bwd: fp32 logits [B,D,h,w] expected")
    B, D, h, w = logits.shape
    for g in (grad_depth, grad_index, grad_entropy):
        if g is not None and (g.dtype != torch.float32 or tuple(g.shape) != (B, h, w)):
            raise ValueError("pscv.softargmin_bwd: upstream gradients must be fp32 [B,h,w]")
    if grad_depth is not None and (depth is None or depth.dtype != torch.float32 or depth.shape[:2] != (B, D)):
        raise ValueError("pscv.softargmin_bwd: grad_depth needs fp32 depth planes [B,D] or [B,D,h,w]")
    out = torch.empty((B, D, h, w, 8), dtype=dtype, device=logits.device)
    rc = _launch("softargmin_bwd", lambda: L.lib().pscv_softargmin_bwd(
        _p(logits), _p(depth), 0 if depth is None else depth.stride(0), int(depth is not None and depth.dim() == 4), _p(grad_depth),
        _p(grad_index), _p(grad_entropy), _p(out), _TORCH2PSCV[dtype], B, D, h, w, _stream()))
    L.check(rc, "pscv_softargmin_bwd")
    return out


def relu_bwd_sum(dout: torch.Tensor, out: torch.Tensor, slope: float = 0.0):
    """``relu_bwd`` that also returns the per-channel sum of the stored result (fp32 [C]): the bias gradient of a conv + bias +
    (Leaky)ReLU layer in the same pass (pscv_leaky_relu_bwd_sum)."""
    _dev(dout, out)
    _vol16(out, "relu_bwd_sum")
    if dout.shape != out.shape or dout.dtype != out.dtype:
        raise ValueError("pscv.relu_bwd_sum: dout must match out")
    Cc = out.shape[4]
    dpre = torch.empty_like(out)
    sums = torch.empty((2, Cc), dtype=torch.float32, device=out.device)
    ws = _workspace(out.device, 0)
    rc = _launch("relu_bwd_sum", lambda: L.lib().pscv_leaky_relu_bwd_sum(_p(dout), _p(out), _dt(out), out.numel() // Cc, Cc, float(slope), _p(dpre),
                                                                         _p(ws), _p(sums), _stream()))
    L.check(rc, "pscv_leaky_relu_bwd_sum")
    return dpre, sums[0]


def relu_bwd(dout: torch.Tensor, out: torch.Tensor, slope: float = 0.0) -> torch.Tensor:
    """dout * [out > 0] on 16-bit channels-last volumes (pscv_relu_bwd): backward of a ReLU applied after a residual add.
    ``slope`` > 0: dout * (out > 0 ? 1 : slope), the backward of LeakyReLU(slope) from its OUTPUT (pscv_leaky_relu_bwd)."""
    _dev(dout, out)
    _vol16(out, "relu_bwd")
    if dout.shape != out.shape or dout.dtype != out.dtype:
        raise ValueError("pscv.relu_bwd: dout must match out")
    Cc = out.shape[4]
    dpre = torch.empty_like(out)
    rc = _launch("relu_bwd", lambda: L.lib().pscv_leaky_relu_bwd(_p(dout), _p(out), _dt(out), out.numel() // Cc, Cc, float(slope), _p(dpre), _stream()))
    L.check(rc, "pscv_leaky_relu_bwd")
    return dpre


def fuse_pairs_bwd(interms: Sequence[torch.Tensor], uncerts: Sequence[torch.Tensor], grad_fused: torch.Tensor):
    """Backward of ``fuse_pairs``: ([d interm_v] 16-bit, [d uncert_v] fp32 [B,h,w]) (pscv_fuse_pairs_bwd)."""
    interms, uncerts = list(interms), list(uncerts)
    _dev(grad_fused, *interms, *uncerts)
    B, D, h, w, c = interms[0].shape
    if grad_fused.shape != interms[0].shape or grad_fused.dtype != interms[0].dtype:
        raise ValueError("pscv.fuse_pairs_bwd: grad_fused must match the pair volumes")
    n = len(interms)
    dI = [torch.empty_like(t) for t in interms]
    dU = [torch.empty((B, h, w), dtype=torch.float32, device=grad_fused.device) for _ in range(n)]
    ip = (C.c_void_p * n)(*[t.data_ptr() for t in interms])
    up = (C.c_void_p * n)(*[t.data_ptr() for t in uncerts])
    dip = (C.c_void_p * n)(*[t.data_ptr() for t in dI])
    dup = (C.c_void_p * n)(*[t.data_ptr() for t in dU])
    rc = _launch("fuse_pairs_bwd", lambda: L.lib().pscv_fuse_pairs_bwd(ip, up, n, _dt(interms[0]), _p(grad_fused), dip, dup, B, D, h, w,
                                                                      _stream()))
    L.check(rc, "pscv_fuse_pairs_bwd")
    return dI, dU


def conv3d_wgrad(p: torch.Tensor, q: torch.Tensor, *, ca: int, cb: int, stride: int, p_coff: int = 0, q_coff: int = 0) -> torch.Tensor:
    """dw[a][b][tz,ty,tx] = sum P[o,a] Q[stride*o + t - 1, b] -> fp32 [ca,cb,3,3,3] (pscv_conv3d_wgrad).
    Conv3d: p = grad of the output, q = input; ConvTranspose3d: p = input, q = grad of the output."""
    _dev(p, q)
    _vol16(p, "conv3d_wgrad")
    _vol16(q, "conv3d_wgrad")
    B, Dp, Hp, Wp, pcs = p.shape
    if q.dtype != p.dtype or tuple(q.shape[:4]) != (B, stride * Dp, stride * Hp, stride * Wp):
        raise ValueError(f"pscv.conv3d_wgrad: q must be [B,{stride}*Dp,{stride}*Hp,{stride}*Wp,*] in p's dtype")
    n = L.lib().pscv_conv3d_wgrad_workspace(B, Dp, Hp, Wp, ca, cb, stride)
    if n < 0:
        L.check(int(n), "pscv_conv3d_wgrad_workspace")
    ws = _workspace(p.device, n)
    dw = torch.empty((ca, cb, 3, 3, 3), dtype=torch.float32, device=p.device)
    rc = _launch(f"conv3d_wgrad[{ca}x{cb},s{stride}]", lambda: L.lib().pscv_conv3d_wgrad(
        _p(p), pcs, p_coff, ca, _p(q), q.shape[4], q_coff, cb, _dt(p), B, Dp, Hp, Wp, stride, _p(ws), _p(dw), 0, _stream()))
    L.check(rc, "pscv_conv3d_wgrad")
    return dw


def warp_cost_bwd(ref: Optional[torch.Tensor], srcs: Sequence[torch.Tensor], cams: torch.Tensor, depth: torch.Tensor,
                  grad_out: torch.Tensor, *, geom: int = L.GEOM_PROJ, cost: int = L.COST_VARIANCE, temp: float = 0.0,
                  ref_hw: Optional[Sequence[int]] = None, want_dtemp: bool = False):
    """Backward of ``warp_cost`` to the feature maps: returns (dref fp32 [B,h,w,C] or None, [dsrc fp32 [B,hs,ws,C]], dtemp
    fp32 [1] or None).  ``grad_out`` has the layout of the forward's output (16-bit like the features, or fp32)."""
    srcs = list(srcs)
    _dev(ref, cams, depth, grad_out, *srcs)
    B, hs, ws, Cc = srcs[0].shape
    h, w = (ref.shape[1:3] if ref is not None else ((hs, ws) if ref_hw is None else (int(ref_hw[0]), int(ref_hw[1]))))
    n = len(srcs)
    D = depth.shape[1]
    per_pixel = depth.dim() == 4
    if cost == L.COST_GROUPCORR:
        shape = (n, B, D, h, w, Cc // 4)
    elif cost == L.COST_WARP_ONLY:
        shape = (n, B, D, h, w, Cc)
    else:
        shape = (B, D, h, w, Cc)
    if tuple(grad_out.shape) != shape:
        raise ValueError(f"pscv.warp_cost_bwd: grad_out has shape {tuple(grad_out.shape)}, expected {shape}")
    dref = torch.zeros((B, h, w, Cc), dtype=torch.float32, device=srcs[0].device) if ref is not None else None
    dsrcs = [torch.zeros((B, hs, ws, Cc), dtype=torch.float32, device=srcs[0].device) for _ in srcs]
    dtemp = torch.zeros((1,), dtype=torch.float32, device=srcs[0].device) if want_dtemp else None
    sp = (C.c_void_p * n)(*[s.data_ptr() for s in srcs])
    dp = (C.c_void_p * n)(*[s.data_ptr() for s in dsrcs])
    rc = _launch(f"warp_cost_bwd[{cost}]", lambda: L.lib().pscv_warp_cost_bwd(
        _p(ref), sp, n, _p(cams), _p(depth), depth.stride(0), int(per_pixel), geom, cost, float(temp), _p(grad_out), _p(dref), dp,
        _p(dtemp), B, Cc, h, w, hs, ws, D, _dt(srcs[0]), _dt(grad_out), _stream()))
    L.check(rc, "pscv_warp_cost_bwd")
    return dref, dsrcs, dtemp
