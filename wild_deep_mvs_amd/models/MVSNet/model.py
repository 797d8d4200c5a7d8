"""Drop-in for the reference's ``models/MVSNet/model.py`` (MVSNet / MVSNet-s) on the pscv HIP engine.

Same constructor, ``forward(imgs, K, R, t, depth_min, depth_max, reference_frame=0, **kwargs)`` signature,
return dict and state-dict key names/shapes as the reference (models/MVSNet/model.py:87-218), so released
checkpoints load unchanged.  Everything from the images to depth + confidence runs as HIP launches:

    8 MFMA conv2d launches (2-D FeatureNet, all views batched; ``feature_engine = "torch"`` keeps PyTorch-ROCm)  ->
    fused warp + variance/softmin  ->  11 MFMA conv3d launches (BN/ReLU/skip fused)  ->  fused softargmin
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import os

import torch
import torch.nn as nn

from ... import _lib as L
from ... import ops
from ... import training as T
from ...graph import ReplayHooks, replayable
from .module import ConvBnReLU, ConvBnReLU3D, deconv_engine_layer, homo_warping, depth_regression  # noqa: F401


def build_proj_matrices(K: torch.Tensor, R: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """[[K R, K t], [0 0 0 1]] (reference utils/utils_3D.py:50-62; A0 of the path, stays torch)."""
    P = torch.zeros(K.shape[:-2] + (4, 4), device=K.device, dtype=K.dtype)
    P[..., :3, :3] = K @ R
    P[..., :3, 3:] = K @ t
    P[..., 3, 3] = 1
    return P


class FeatureNet(nn.Module):
    """Upstream 2-D extractor, [B,3,H,W] -> [B,32,H/4,W/4] (reference models/MVSNet/model.py:21-41).

    ``forward`` is the plain PyTorch module (training, CPU tools).  ``forward_engine`` runs the same eight layers as
    eight MFMA ``pscv_conv2d`` launches on channels-last 16-bit maps (BatchNorm folded, ReLU fused) and returns the
    [B,h,w,32] map the warp kernel reads -- SURVEY section 8f-2."""

    SPEC = [(3, 8, 3, 1, 1), (8, 8, 3, 1, 1), (8, 16, 5, 2, 2), (16, 16, 3, 1, 1), (16, 16, 3, 1, 1),
            (16, 32, 5, 2, 2), (32, 32, 3, 1, 1)]

    def __init__(self):
        super().__init__()
        self.inplanes = 32
        for i, (ci, co, k, s, p) in enumerate(self.SPEC):
            setattr(self, f"conv{i}", ConvBnReLU(ci, co, k, s, p))
        self.feature = nn.Conv2d(32, 32, 3, 1, 1)
        self._layers = None
        self._layers_key = None

    def forward(self, x):
        for i in range(7):
            x = getattr(self, f"conv{i}")(x)
        return self.feature(x)

    def engine_layers(self, dtype: torch.dtype):
        """Packed weights + folded BatchNorm of the eight layers, rebuilt when a parameter / buffer changes."""
        key = (ops.weights_epoch(), dtype) + tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))
        if self._layers is None or self._layers_key != key:
            dev = self.feature.weight.device
            layers = []
            for i, (ci, co, k, s, p) in enumerate(self.SPEC):
                blk = getattr(self, f"conv{i}")
                bn = blk.bn
                layers.append(ops.Conv2dLayer.build(blk.conv.weight, stride=s, device=dev, relu=True, dtype=dtype, bn_eps=bn.eps,
                                                    bn=(bn.weight, bn.bias, bn.running_mean, bn.running_var)))
            layers.append(ops.Conv2dLayer.build(self.feature.weight, stride=1, device=dev, conv_bias=self.feature.bias,
                                                relu=False, dtype=dtype))
            self._layers, self._layers_key = layers, key
        return self._layers

    def forward_engine(self, x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        """[B,3,H,W] image batch on the GPU -> channels-last features [B,H/4,W/4,32] in ``dtype`` (eval mode)."""
        if self.training:
            raise NotImplementedError("pscv FeatureNet: the HIP path is inference-only (eval-mode BatchNorm is folded)")
        y = ops.image_to_channels_last8(x, dtype)
        for layer in self.engine_layers(dtype):
            y = ops.conv2d(y, layer)
        return y


def _deconv_block(ci: int, co: int) -> nn.Sequential:
    return nn.Sequential(nn.ConvTranspose3d(ci, co, kernel_size=3, padding=1, output_padding=1, stride=2, bias=False),
                         nn.BatchNorm3d(co), nn.ReLU(inplace=True))


# prob head + softmax regression as ONE fused tail (ops.prob_softargmin: the head's depth sweep keeps running softmax statistics and a
# merge launch finishes them).  Built, parity-tested (tests/test_gpu_conv3d.py, tests/test_gpu_mvsnet.py) and measured SLOWER than the two
# separate launches at the headline size: 52.5 us against 26.0 + 13.5 us (same box, both at three waves per SIMD).  The head's sweep is instruction-issue bound (three resident
# workgroups per CU, ~250 instructions per 6-plane block in one dependency chain); the ~100 extra vector instructions per block of
# the online softmax lengthen exactly that chain, while the stand-alone softargmin pass is latency bound with an idle vector ALU.
# So the default stays off.
FUSED_TAIL = False
# conv11^T (+ BatchNorm, ReLU, skip) and the prob head as ONE depth sweep (ops.tail_sweep, csrc/conv3d_tail.hip; round 5): the
# full-resolution 8-channel volume between them is neither written nor read; the logits are bit-identical to the two launches.
# `taps` (tests, diagnostics) keep the two launches so that the intermediate volume exists.
TAIL_SWEEP = True
# ... and the softmax regression folded into that sweep (per-chunk statistics + one merge launch instead of the softargmin pass): built,
# parity-tested (tests/test_gpu_conv3d.py: depth within 2e-6 of the range) and measured at the headline size: sweep + merge 62.4 us
# against 44.7 + 12.9 us for the sweep and the stand-alone softargmin pass (the ~100 extra vector instructions per 6-plane block sit
# in the consume phase's dependency chain, as with the round-2 prob + softargmin fusion) -> off by default.
TAIL_SWEEP_REGRESS = False


class CostRegNet(nn.Module):
    """3-D U-Net regulariser (reference models/MVSNet/model.py:43-84) on MFMA conv3d launches.

    The sub-modules only hold parameters under the reference's names; ``forward`` takes and returns the
    engine's channels-last 16-bit volumes: [B,D,h,w,32] -> fp32 logits [B,D,h,w]."""

    def __init__(self):
        super().__init__()
        self.conv0 = ConvBnReLU3D(32, 8)
        self.conv1 = ConvBnReLU3D(8, 16, stride=2)
        self.conv2 = ConvBnReLU3D(16, 16)
        self.conv3 = ConvBnReLU3D(16, 32, stride=2)
        self.conv4 = ConvBnReLU3D(32, 32)
        self.conv5 = ConvBnReLU3D(32, 64, stride=2)
        self.conv6 = ConvBnReLU3D(64, 64)
        self.conv7 = _deconv_block(64, 32)
        self.conv9 = _deconv_block(32, 16)
        self.conv11 = _deconv_block(16, 8)
        self.prob = nn.Conv3d(8, 1, 3, stride=1, padding=1)
        self._layers: Optional[Dict[str, ops.Conv3dLayer]] = None
        self._layers_key = None

    # -- weight residency ------------------------------------------------------------------
    def _param_key(self):
        return (ops.weights_epoch(),) + tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))

    def engine_layers(self, dtype: torch.dtype) -> Dict[str, ops.Conv3dLayer]:
        """Packed 16-bit weights + folded eval-mode BN, rebuilt whenever a parameter changed
        (load_state_dict, .to(), optimizer step) or the storage format changed."""
        key = (dtype,) + self._param_key()
        if self._layers is None or key != self._layers_key:
            dev = self.prob.weight.device
            lay = {f"conv{i}": getattr(self, f"conv{i}").engine_layer(dev, dtype) for i in range(7)}
            for n in ("conv7", "conv9", "conv11"):
                lay[n] = deconv_engine_layer(getattr(self, n), dev, dtype=dtype)
            lay["prob"] = ops.Conv3dLayer.build(self.prob.weight, kind=L.CONV_S1, device=dev, conv_bias=self.prob.bias,
                                                dtype=dtype)
            self._layers, self._layers_key = lay, key
        return self._layers

    def train_blocks(self) -> List[T.Block]:
        """The U-Net as the block list of the training executor (``training.RegressFn``): same dataflow as ``forward``
        (reference models/MVSNet/model.py:74-84), BatchNorm with batch statistics."""
        blocks, prev = [], "cost"
        for i, stride in enumerate((1, 2, 1, 2, 1, 2, 1)):
            m = getattr(self, f"conv{i}")
            blocks.append(T.Block(f"conv{i}", prev, m.conv.weight, stride=stride, bn=m.bn, relu=True))
            prev = f"conv{i}"
        for name, skip in (("conv7", "conv4"), ("conv9", "conv2"), ("conv11", "conv0")):
            seq = getattr(self, name)
            blocks.append(T.Block(name, prev, seq[0].weight, stride=2, transposed=True, bn=seq[1], relu=True, skip=skip))
            prev = name
        blocks.append(T.Block("prob", prev, self.prob.weight, bn=None, relu=False, conv_bias=self.prob.bias))
        return blocks

    def forward(self, cost: torch.Tensor, taps: Optional[dict] = None, regress: Optional[torch.Tensor] = None):
        """cost [B,D,h,w,32] -> fp32 logits [B,D,h,w].  With ``regress`` = the per-batch depth planes [B,D], the tail runs fused
        (``ops.prob_softargmin``: the prob head emits softmax partials, one merge launch gives depth and confidence) and the
        return value is (logits, {"depth", "conf"}); sizes the fused head does not take fall back to the two separate launches."""
        if self.training:
            raise RuntimeError("pscv CostRegNet: in train() mode the U-Net runs inside training.RegressFn (MVSNet.forward "
                               "routes there); this entry point is the eval-mode engine")
        B, D, h, w, _ = cost.shape
        if D % 8 or h % 8 or w % 8:
            raise ValueError(f"MVSNet CostRegNet needs D,h,w multiples of 8 (got {D},{h},{w}), as in the reference")
        ly = self.engine_layers(cost.dtype)
        c0 = ops.conv3d(cost, ly["conv0"])
        c2 = ops.conv3d(ops.conv3d(c0, ly["conv1"]), ly["conv2"])
        c4 = ops.conv3d(ops.conv3d(c2, ly["conv3"]), ly["conv4"])
        c6 = ops.conv3d(ops.conv3d(c4, ly["conv5"]), ly["conv6"])
        u7 = ops.conv3d(c6, ly["conv7"], skip=c4)      # conv4 + relu(bn(deconv))     model.py:79
        u9 = ops.conv3d(u7, ly["conv9"], skip=c2)      # model.py:80
        fused = u11 = logits = None
        if TAIL_SWEEP and taps is None:                    # model.py:81-82 (+ 207-215 with the regression fused: TAIL_SWEEP_REGRESS)
            r = ops.tail_sweep(u9, ly["conv11"], ly["prob"], skip=c0, regress=regress if TAIL_SWEEP_REGRESS else None)
            if isinstance(r, dict):
                fused, logits = r, r["logits"]
            else:
                logits = r
        if logits is None:
            u11 = ops.conv3d(u9, ly["conv11"], skip=c0)    # model.py:81
            fused = ops.prob_softargmin(u11, ly["prob"], regress) if regress is not None and FUSED_TAIL else None
            logits = fused["logits"] if fused is not None else ops.conv3d(u11, ly["prob"], out_dtype=torch.float32).view(B, D, h, w)
        if taps is not None:
            taps.update(conv0=c0, conv2=c2, conv4=c4, conv6=c6, up7=u7, up9=u9, up11=u11)
        if regress is None:
            return logits
        if fused is None:
            fused = ops.softargmin(logits, regress, want_conf=True, conf_mode=0)
        return logits, {"depth": fused["depth"], "conf": fused["conf"]}


def _regnet_forward_depth_shard(self, x_ext: torch.Tensor, group) -> torch.Tensor:
    """``CostRegNet`` on one rank's depth planes (the recommended shard for MVSNet / CVP; reference
    models/MVSNet/model.py:43-84 has no counterpart).  ``x_ext`` [1, n + 4, h, w, 32]: the n owned planes (a multiple of 8) at
    [2, n + 2), valid cost planes in the inner halo slots (the warp kernel computes them itself: no exchange for the first
    layer), zeros beyond the ends of the volume.  Every layer runs the single-GPU kernel on the extended tensor -- the U-Net's
    receptive field along depth is +-30 planes, so the halo cannot be recomputed like Vis-MVSNet's: ONE boundary plane per
    layer and neighbour is exchanged instead (``dist.halo_sync``, 10 exchanges per forward) -- and keeps the owned planes at
    [2, n_level + 2) of its output:
      stride 1: out_ext = conv(in_ext);  stride 2: output plane t of the rank reads local input planes 2t+1 .. 2t+3, i.e. it is
      output t + 1 of conv(in_ext) -> written at out_ext[1:];  transposed stride 2: run on in_ext[1 : m + 3] (halo, owned, halo),
      whose 2 (m + 2) output planes are exactly out_ext.
    Returns fp32 logits [1, n + 4, h, w] (owned planes at [2, n + 2))."""
    from ... import dist as pdist
    if self.training:
        raise RuntimeError("pscv CostRegNet: the depth-plane shard is an inference path")
    B, ne, h, w, _ = x_ext.shape
    n = ne - 4
    if B != 1 or n % 8 or h % 8 or w % 8:
        raise ValueError(f"depth-plane shard: one batch item, owned planes / h / w multiples of 8 (got B={B}, n={n}, {h}x{w})")
    ly = self.engine_layers(x_ext.dtype)
    dt, dev = x_ext.dtype, x_ext.device
    sync = lambda y: pdist.halo_sync(y, group)

    def down(x, layer, n_out, c_out, hh, ww):          # stride 2: owned output planes land at [2, n_out + 2)
        y = torch.empty((1, n_out + 4, hh, ww, c_out), dtype=dt, device=dev)
        ops.conv3d(x, layer, out=y[:, 1:n_out + 3])
        sync(y)
        return y

    def same(x, layer):
        y = ops.conv3d(x, layer)
        sync(y)
        return y

    def up(x, layer, skip):                            # transposed stride 2 on (halo, owned, halo)
        m = x.shape[1] - 4
        y = ops.conv3d(x[:, 1:m + 3], layer, skip=skip)
        sync(y)
        return y
    c0 = same(x_ext, ly["conv0"])
    c2 = same(down(c0, ly["conv1"], n // 2, 16, h // 2, w // 2), ly["conv2"])
    c4 = same(down(c2, ly["conv3"], n // 4, 32, h // 4, w // 4), ly["conv4"])
    c6 = same(down(c4, ly["conv5"], n // 8, 64, h // 8, w // 8), ly["conv6"])
    u7 = up(c6, ly["conv7"], c4)
    u9 = up(u7, ly["conv9"], c2)
    u11 = up(u9, ly["conv11"], c0)
    return ops.conv3d(u11, ly["prob"], out_dtype=torch.float32).view(1, ne, h, w)


CostRegNet.forward_depth_shard = _regnet_forward_depth_shard


class MVSNet(ReplayHooks, nn.Module):
    def __init__(self, aggregation="variance"):
        super().__init__()
        if aggregation not in ("variance", "softmin"):
            raise NotImplementedError("Aggregation: " + aggregation)
        self.feature = FeatureNet()
        self.cost_regularization = CostRegNet()
        if aggregation == "softmin":
            self.register_parameter("temp", torch.nn.Parameter(torch.ones((1))))   # model.py:94-95
        self.aggregation = aggregation
        self.num_depth = 192
        # HBM storage format of features / cost volume / activations (arithmetic is fp32 either way).
        # fp16 (11-bit significand) keeps depth within ~2e-4 relative L1 of the fp32 reference; bf16 (8-bit)
        # sits at ~1e-3 on peaked-but-unsaturated softmaxes, so fp16 is the default.
        self.storage_dtype = torch.float16
        # 2-D extractor: "pscv" = eight MFMA conv2d launches writing the warp kernel's layout directly (16-bit
        # activations between the layers); "torch" = PyTorch-ROCm in fp32, converted once at the end.
        self.feature_engine = "pscv"
        # train() mode: gradients span many decades, so the stored activations / gradients default to bf16 there
        self.train_storage_dtype = torch.bfloat16
        # 2-D extractor in train(): "torch" (default) = PyTorch-ROCm autograd in fp32, the reference's numerics for the step
        # before the path; "pscv" = training.FeatureNetFn (forward and backward on the engine's conv2d / weight-gradient /
        # BatchNorm kernels with 16-bit stored activations: a mixed-precision speed mode, eight more 16-bit roundings in front
        # of the sweep -- depth moves by ~2e-3 (fp16) / ~2e-2 (bf16) relative on the training fixtures)
        self.feature_engine_train_dtype = None       # 16-bit format of the engine extractor's activations in train() (None: fp16)
        self.feature_engine_train = os.environ.get("PSCV_FEATURE_ENGINE_TRAIN", "torch")
        # source-view shard: with a torch.distributed group set here (``set_view_group``), rank r warps
        # source views r, r+G, ... only and contributes fp32 partial sums (sum f, sum f^2; rank 0 adds the reference view);
        # one all-reduce (RCCL) and pscv_variance_finish give every rank the full cost volume.  The reference has no
        # counterpart.  (At the headline size the all-reduce of 2 x 503 MB costs more than warping all views locally --
        # DESIGN.md section 7 -- it pays off only when the per-view work dominates: many views, large maps.)
        self.view_group = None
        # depth-plane shard (``set_depth_group``): rank r sweeps, regularises and regresses planes
        # [a, b) of every reference view: the warp needs nothing from the other ranks, each of the 11 U-Net layers exchanges one
        # boundary plane with each neighbour, the softmax over D is merged from log-sum-exp partials.  Strong scaling of ONE
        # reference view; pays off when a rank's share of the regulariser outweighs ~10 point-to-point latencies
        self.depth_group = None
        # batch items of the eval-mode hot path on separate HIP streams (``_hot_path_streams``): off by default (a caller's batch runs as
        # one launch per layer on the caller's stream); bench.py switches it on.  Every kernel is bit-stable under that overlap since
        # round 4 (tests/test_gpu_overlap.py).  ``batch_streams_capture``: under a hipGraph capture the fork becomes parallel branches
        # of the graph -- correct on replay with changing inputs since the warp kernel ships as its scalar build (round 3 saw wrong
        # replays and blamed ROCm; it was the packed build's overlap defect), 3 % faster than the batched graph at three views.
        self.batch_streams_capture = True
        self.batch_streams = False
        # stream mode, staggered (round 5 experiment, OFF): item b + 1's warp + cost launch waits for item b's (an event edge between
        # the streams; a graph edge under capture), so that the items do not run in lockstep -- three warps, then three conv0s ... --
        # but one item's warp beside another's conv0 / tail.  Measured at the headline size, alternating: 0.977 / 0.975 ms staggered
        # against 0.949 / 0.955 ms in lockstep: a warp launch fills every CU's LDS (4 x 40 KB), so another item's conv0 (70 KB per
        # workgroup) cannot move in beside it anyway, and the stagger only delays the later items.
        self.batch_stagger = False

    # -- upstream ---------------------------------------------------------------------------
    def extract_features(self, imgs: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        """All views through the PyTorch 2-D extractor as ONE batch (eval-mode BatchNorm is a per-channel affine, so
        the result equals the reference's per-view loop, model.py:101-107).  NCHW fp32 maps."""
        imgs = list(imgs)
        if len({tuple(i.shape) for i in imgs}) == 1:
            return list(torch.chunk(self.feature(torch.cat(imgs, 0)), len(imgs), 0))
        return [self.feature(img) for img in imgs]

    def extract_features_cl(self, imgs: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        """Channels-last 16-bit feature maps [B,h,w,32] of all views: the HIP extractor (``feature_engine = "pscv"``,
        default) or PyTorch-ROCm followed by a layout / precision conversion (``"torch"``)."""
        imgs = list(imgs)
        if self.feature_engine == "pscv":
            if len({tuple(i.shape) for i in imgs}) == 1:
                f = self.feature.forward_engine(ops.batch_views(imgs), self.storage_dtype)
                return list(torch.chunk(f, len(imgs), 0))
            return [self.feature.forward_engine(img, self.storage_dtype) for img in imgs]
        return [ops.to_channels_last(f, self.storage_dtype) for f in self.extract_features(imgs)]

    def set_view_group(self, group):
        """Shard the source views of the variance cost volume over a torch.distributed group (None = no sharding)."""
        self.view_group = group

    def set_depth_group(self, group):
        """Shard the depth planes of the whole hot path over a torch.distributed group (None = no sharding)."""
        self.depth_group = group

    def _hot_path_depth_shard(self, features_cl, proj, depth_values, reference_frame):
        import torch.distributed as dist
        from ... import dist as pdist
        grp = self.depth_group
        world, rank = dist.get_world_size(grp), dist.get_rank(grp)
        V = len(features_cl)
        src_idx = [i for i in range(V) if i != reference_frame]
        B, h, w, C = features_cl[reference_frame].shape
        D = depth_values.shape[1]
        if D % 8 or h % 8 or w % 8:
            raise ValueError(f"MVSNet CostRegNet needs D,h,w multiples of 8 (got {D},{h},{w}), as in the reference")
        a, b = pdist.plane_shard(D, world, rank, multiple=8)
        n = b - a
        if n == 0:
            raise ValueError(f"depth-plane shard: {world} ranks for {D} planes leaves rank {rank} without an 8-plane block")
        cams = ops.proj_cams_device(proj.to(torch.float32).contiguous(), reference_frame)
        dv = depth_values.to(torch.float32)
        lo, hi = max(0, a - 2), min(D, b + 2)              # the planes of the extended range [a - 2, b + 2) that exist
        code = L.COST_VARIANCE if self.aggregation == "variance" else L.COST_SOFTMIN
        if code == L.COST_SOFTMIN:
            key = (ops.weights_epoch(), self.temp.data_ptr(), self.temp._version)
            if getattr(self, "_temp_key", None) != key:
                self._temp_val, self._temp_key = float(self.temp.detach().float().item()), key
        depths, confs = [], []
        for bi in range(B):                                # (plane slices of one batch item are contiguous views)
            fb = [f[bi:bi + 1] for f in features_cl]
            cost_ext = torch.empty((1, n + 4, h, w, C), dtype=fb[0].dtype, device=fb[0].device)
            cost_ext[:, :lo - (a - 2)].zero_()
            cost_ext[:, hi - (a - 2):].zero_()
            ops.warp_cost(fb[reference_frame], [fb[i] for i in src_idx], cams[:, bi:bi + 1].contiguous(), dv[bi:bi + 1, lo:hi].contiguous(),
                          geom=L.GEOM_PROJ, cost=code, temp=getattr(self, "_temp_val", 0.0), out=cost_ext[:, lo - (a - 2):hi - (a - 2)])
            logits = self.cost_regularization.forward_depth_shard(cost_ext, grp)[:, 2:n + 2].contiguous()
            part = ops.softargmin(logits, dv[bi:bi + 1, a:b].contiguous(), want_partials=True, index_offset=a)["partials"]
            depth, index, m, Z = pdist.merge_partials_stats(part, grp)
            depths.append(depth)
            confs.append(pdist.photometric_confidence_shard(logits, m, Z, index, a, grp))
        return torch.cat(depths, 0), torch.cat(confs, 0)

    def _sharded_variance(self, ref_feature, src_features, cams, depth_values):
        import torch.distributed as dist
        world, rank = dist.get_world_size(self.view_group), dist.get_rank(self.view_group)
        mine = [i for i in range(len(src_features)) if i % world == rank]
        B, h, w, C = ref_feature.shape
        if mine:
            sums = ops.warp_cost(ref_feature if rank == 0 else None, [src_features[i] for i in mine], cams[mine].contiguous(),
                                 depth_values, geom=L.GEOM_PROJ, cost=L.COST_VARIANCE_PARTIAL, ref_hw=(h, w))
        else:   # more ranks than source views: this rank only joins the collective
            sums = torch.zeros((2, B, depth_values.shape[1], h, w, C), dtype=torch.float32, device=ref_feature.device)
        dist.all_reduce(sums, group=self.view_group)
        return ops.variance_finish(sums, len(src_features) + 1, cost=L.COST_VARIANCE, dtype=ref_feature.dtype)

    # -- hot path ---------------------------------------------------------------------------
    def build_cost_volume(self, ref_feature, src_features, ref_proj, src_projs, depth_values, cams=None):
        """Channels-last features [B,h,w,32] + [B,4,4] projections + planes [B,D] (or [B,D,h,w])
        -> channels-last cost volume [B,D,h,w,32] in one fused launch (reference model.py:109-176)."""
        if cams is None:
            cams = ops.proj_cams_device(torch.stack([ref_proj] + list(src_projs), dim=1).to(torch.float32).contiguous(), 0)
        if self.view_group is not None:
            if self.aggregation != "variance":
                raise NotImplementedError("pscv MVSNet: the source-view shard reduces variance sums; soft-min weights need all views")
            return self._sharded_variance(ref_feature, list(src_features), cams, depth_values)
        if self.aggregation == "variance":
            return ops.warp_cost(ref_feature, src_features, cams, depth_values, geom=L.GEOM_PROJ,
                                 cost=L.COST_VARIANCE, out_dtype=ref_feature.dtype)
        # the kernel takes the temperature by value: read it back once per parameter version (no host sync in steady state,
        # and none inside a hipGraph capture of the forward)
        key = (ops.weights_epoch(), self.temp.data_ptr(), self.temp._version)
        if getattr(self, "_temp_key", None) != key:
            self._temp_val, self._temp_key = float(self.temp.detach().float().item()), key
        return ops.warp_cost(ref_feature, src_features, cams, depth_values, geom=L.GEOM_PROJ, cost=L.COST_SOFTMIN,
                             temp=self._temp_val, out_dtype=ref_feature.dtype)

    def hot_path(self, features_cl: Sequence[torch.Tensor], proj: torch.Tensor, depth_values: torch.Tensor,
                 reference_frame: int = 0, taps: Optional[dict] = None, _warp_gate=None):
        """features_cl: V channels-last maps [B,h,w,32]; proj [B,V,4,4]; depth_values [B,D] fp32 (reference view).
        Returns (depth [B,h,w], photometric_confidence [B,h,w]) -- reference model.py:197-215.
        ``_warp_gate`` = (event to wait for, event to record) around the warp + cost launch: the staggered stream mode's edges
        between the views of a batch (``_hot_path_streams``); an argument, not instance state, so that the call is re-entrant."""
        if self.depth_group is not None:
            if self.view_group is not None or taps is not None:
                raise NotImplementedError("pscv MVSNet: the depth-plane shard excludes the source-view shard and taps")
            return self._hot_path_depth_shard(features_cl, proj, depth_values, reference_frame)
        B = features_cl[0].shape[0]
        if (2 <= B <= self.MAX_BATCH_STREAMS and self.batch_streams and taps is None and self.view_group is None and features_cl[0].is_cuda
                and (not torch.cuda.is_current_stream_capturing() or self.batch_streams_capture)):
            return self._hot_path_streams(features_cl, proj, depth_values, reference_frame)
        V = len(features_cl)
        src_idx = [i for i in range(V) if i != reference_frame]
        cams = ops.proj_cams_device(proj.to(torch.float32).contiguous(), reference_frame)
        gate = _warp_gate                                  # (wait for, record) events of the staggered stream mode, else None
        if gate is not None and gate[0] is not None:
            torch.cuda.current_stream().wait_event(gate[0])
        cost = self.build_cost_volume(features_cl[reference_frame], [features_cl[i] for i in src_idx],
                                      proj[:, reference_frame], [proj[:, i] for i in src_idx], depth_values, cams)
        if gate is not None and gate[1] is not None:
            gate[1].record(torch.cuda.current_stream())
        logits, o = self.cost_regularization(cost, taps, regress=depth_values.to(torch.float32).contiguous())
        if taps is not None:
            taps.update(cost_volume=cost, logits=logits)
        return o["depth"], o["conf"]

    MAX_BATCH_STREAMS = 4

    def _hot_path_streams(self, features_cl, proj, depth_values, reference_frame):
        """The reference views of a batch are independent objects whose hot paths have COMPLEMENTARY bottlenecks: the warp is
        bound by vector-ALU issue, conv0 and the full-resolution layers by the matrix cores and the CUs' memory path.  Batch
        item b therefore runs on its own HIP stream (fork from / join into the caller's stream), so that one item's warp shares
        the chip with another item's U-Net: measured at the headline size with eager launches, two views 0.673-0.693 ms against
        0.725-0.740 ms for the batched launches on one stream, three views ~0.33 ms per view (`scripts/dev/graph_branch_toy.py`,
        `bench.py`), outputs bit-equal to the one-item runs (tests/test_gpu_mvsnet.py).  Under a hipGraph capture the items become
        parallel branches of the graph (``batch_streams_capture``): round 3 saw such graphs replay WRONGLY once inputs changed and blamed
        ROCm 7.2; the cause was the overlap defect of the packed warp build.  With the scalar build 12 of 12
        replays on changing inputs equal the one-item runs bit for bit (tests/test_gpu_mvsnet.py)
        and the forked graph is the fastest form of the three-view step (0.973 ms against 1.007 batched, 1.075 eager streams).
        At most MAX_BATCH_STREAMS items; `net.batch_streams = False` turns it off."""
        B = features_cl[0].shape[0]
        dev = features_cl[0].device
        pool = self.__dict__.setdefault("_side_streams", {})
        streams = pool.get(dev)
        if streams is None or len(streams) < B:
            streams = pool[dev] = [torch.cuda.Stream(device=dev) for _ in range(self.MAX_BATCH_STREAMS)]
        main = torch.cuda.current_stream(dev)
        depth_values = depth_values.to(torch.float32)
        outs = []
        # one item's warp overlaps another item's conv kernels here; every kernel of the engine is bit-stable under that overlap
        # (tests/test_gpu_overlap.py; the LDS-staged warp kernel ships as its scalar-fp32 build for this reason)
        prev_ev = None
        for b in range(B):
            st = streams[b]
            st.wait_stream(main)
            with torch.cuda.stream(st):
                fb = [f[b:b + 1] for f in features_cl]                      # contiguous views of one batch item
                ev = torch.cuda.Event() if (self.batch_stagger and b + 1 < B) else None
                outs.append(self.hot_path(fb, proj[b:b + 1], depth_values[b:b + 1].contiguous(), reference_frame,    # (B = 1: the plain path)
                                          _warp_gate=(prev_ev, ev) if self.batch_stagger else None))
                prev_ev = ev
        for st in streams[:B]:
            main.wait_stream(st)
        return torch.cat([o[0] for o in outs], 0), torch.cat([o[1] for o in outs], 0)

    def hot_path_train(self, features: Sequence[torch.Tensor], proj: torch.Tensor, depth_values: torch.Tensor,
                       reference_frame: int = 0):
        """train()-mode hot path with autograd: NCHW fp32 feature maps (requiring grad) -> depth, confidence.
        Forward and backward are HIP launches (training.WarpCostFn / training.RegressFn); BatchNorm3d uses and
        updates batch statistics like the reference's modules in train() (models/MVSNet/model.py:109-139,74-84)."""
        V = len(features)
        src_idx = [i for i in range(V) if i != reference_frame]
        cl = features[0].dtype == self.train_storage_dtype   # channels-last 16-bit maps from FeatureNetFn (else NCHW fp32)
        D, h, w = (depth_values.shape[1],) + (tuple(features[0].shape[1:3]) if cl else tuple(features[0].shape[2:4]))
        if D % 8 or h % 8 or w % 8:
            raise ValueError(f"MVSNet CostRegNet needs D,h,w multiples of 8 (got {D},{h},{w}), as in the reference")
        dt = self.train_storage_dtype
        cams = ops.proj_cams_device(proj.detach().to(torch.float32).contiguous(), reference_frame)
        cost_mode = L.COST_VARIANCE if self.aggregation == "variance" else L.COST_SOFTMIN
        temp = self.temp if self.aggregation == "softmin" else None
        cost = T.WarpCostFn.apply(cams, depth_values, L.GEOM_PROJ, cost_mode, dt, temp, features[reference_frame],
                                  *[features[i] for i in src_idx])
        blocks = self.cost_regularization.train_blocks()
        return T.RegressFn.apply(blocks, depth_values, dt, cost, *T.RegressFn.block_params(blocks))

    @replayable
    def forward(self, imgs, K, R, t, depth_min, depth_max, reference_frame=0, **kwargs):
        if isinstance(imgs, torch.Tensor):
            imgs = torch.unbind(imgs, 1)
        scaled_K = K.clone()
        scaled_K[:, :, :2] /= 4                                        # model.py:183-184
        proj = build_proj_matrices(scaled_K, R, t)                     # [B,V,4,4]
        if len(imgs) != proj.shape[1]:
            raise AssertionError("Different number of images and projection matrices")
        D = int(self.num_depth)
        steps = torch.arange(D, device=depth_min.device, dtype=torch.float32).view(1, 1, -1)
        depth_values = depth_min.unsqueeze(-1) + ((depth_max - depth_min) / (D - 1)).unsqueeze(-1) * steps  # model.py:187-189
        dv_ref = depth_values[:, reference_frame].to(torch.float32).contiguous()

        if self.training:
            # per-view extractor passes like the reference (model.py:101-107): each view normalises with its own batch
            # statistics; the 2-D extractor is upstream of the path and stays on PyTorch-ROCm autograd in training
            if self.feature_engine_train == "pscv":
                # all views in ONE extractor pass: the views are the group axis of the BatchNorm statistics (each view its own,
                # like the per-view calls of the reference), convolutions and weight gradients run over the whole stack
                fparams = T.FeatureNetFn.params(self.feature)
                stack = torch.cat(list(imgs), 0)
                # the extractor's own 16-bit format: fp16 by default -- its BatchNorm-normalised activations need no bf16 range, and 40 Adam
                # steps on one sample end at the PyTorch-ROCm extractor's loss with fp16 activations (3.24 vs 3.25) but not with bf16
                # ones (3.78 vs 3.37; scripts/dev/train_curve.py); the maps are converted to the sweep's format at the end
                fdt = self.feature_engine_train_dtype or torch.float16
                fs = T.FeatureNetFn.apply(self.feature, fdt, len(imgs), stack, *fparams)
                if fdt != self.train_storage_dtype:
                    fs = fs.to(self.train_storage_dtype)
                feats = list(torch.split(fs, imgs[0].shape[0], 0))
            else:
                feats = [self.feature(img) for img in imgs]
            depth, conf = self.hot_path_train(feats, proj, dv_ref, reference_frame)
            return {"depth": depth, "depth_est_list": [depth, ], "depth_pair_list": [], "photometric_confidence": conf}

        with torch.no_grad():
            feats_cl = self.extract_features_cl(imgs)
            depth, conf = self.hot_path(feats_cl, proj, dv_ref, reference_frame, kwargs.get("taps"))
        return {"depth": depth, "depth_est_list": [depth, ], "depth_pair_list": [],
                "photometric_confidence": conf}
