"""CPU restatement of the MVSNet / MVSNet-s hot path.  TEST INFRASTRUCTURE ONLY.

Functional form: every network stage takes the reference's ``state_dict``
(same key names as ``models/MVSNet/model.py``) instead of ``nn.Module`` objects,
so the same weights drive the reference (golden generation), this oracle and the
HIP engine.  fp32 throughout; BatchNorm in eval mode by default, with the batch statistics
of ``nn.BatchNorm*.train()`` when ``training=True`` (the reference under ``train.py``: gradients
then come from ATen autograd through these same functions, exactly as in the reference).
"""
from __future__ import annotations

from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Mapping[str, torch.Tensor]
BN_EPS = 1e-5  # nn.BatchNorm default, reference models/MVSNet/module.py:44


# --------------------------------------------------------------------------
# A0: cameras and depth planes
# --------------------------------------------------------------------------
def build_proj_matrices(K: torch.Tensor, R: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """[..,3,3],[..,3,3],[..,3,1] -> [..,4,4] = [[K R, K t],[0 0 0 1]].
    Reference ``utils/utils_3D.py:50-62``."""
    P = torch.zeros(K.shape[:-2] + (4, 4), dtype=K.dtype)
    P[..., :3, :3] = K @ R
    P[..., :3, 3:] = K @ t
    P[..., 3, 3] = 1
    return P


def mvsnet_cameras(K, R, t, depth_min, depth_max, num_depth: int):
    """Feature-resolution projection matrices and the per-view depth planes.
    Reference ``models/MVSNet/model.py:183-189``: rows 0-1 of K are divided by 4,
    ``d_i = min + i (max - min) / (D - 1)``."""
    Ks = K.clone()
    Ks[:, :, :2] /= 4
    proj = build_proj_matrices(Ks, R, t)  # [B,V,4,4]
    idx = torch.arange(num_depth).view(1, 1, -1)
    step = (depth_max - depth_min) / (num_depth - 1)
    depth_values = depth_min.unsqueeze(-1) + step.unsqueeze(-1) * idx  # [B,V,D]
    return proj, depth_values


# --------------------------------------------------------------------------
# A1: plane-sweep warp
# --------------------------------------------------------------------------
def sweep_pixel_coords(src_proj, ref_proj, depth_values, ref_hw: Tuple[int, int]):
    """Source-image pixel coordinates of every (plane, reference pixel).

    Follows reference ``models/MVSNet/module.py:127-150``: ``proj = P_src P_ref^-1``;
    ``q = rot [x y 1]^T d + trans`` on integer pixel centres; perspective divide;
    ``q_z <= 0`` sends the sample to (-10, -10).  Returns (u, v, q_z) each
    [B, D, h*w].  ``depth_values`` is [B,D] or [B,D,h,w].
    """
    B = src_proj.shape[0]
    D = depth_values.shape[1]
    h, w = ref_hw
    proj = torch.matmul(src_proj, torch.inverse(ref_proj))
    rot, trans = proj[:, :3, :3], proj[:, :3, 3:4]
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    pix = torch.stack((xx.reshape(-1), yy.reshape(-1), torch.ones(h * w)))  # [3, h*w]
    rot_pix = torch.matmul(rot, pix.unsqueeze(0).expand(B, -1, -1))  # [B,3,h*w]
    dv = depth_values.reshape(B, 1, D, -1)  # [B,1,D,1] or [B,1,D,h*w]
    q = rot_pix.unsqueeze(2) * dv + trans.view(B, 3, 1, 1)  # [B,3,D,h*w]
    uv = q[:, :2] / q[:, 2:3]
    behind = (q[:, 2:3] <= 0).expand(-1, 2, -1, -1)
    uv = torch.where(behind, torch.full_like(uv, -10.0), uv)
    return uv[:, 0], uv[:, 1], q[:, 2]


def homo_warping(src_fea, src_proj, ref_proj, depth_values, ref_shape: Optional[Sequence[int]] = None):
    """[B,C,hs,ws] -> [B,C,D,h,w].  Reference ``models/MVSNet/module.py:111-169``:
    coordinates are normalised by ``(size-1)/2``, clamped to +-10 and handed to
    ``grid_sample(bilinear, zeros, align_corners=True)`` -- i.e. the source is
    sampled at pixel index exactly (u, v)."""
    B, C, hs, ws = src_fea.shape
    h, w = (hs, ws) if ref_shape is None else (int(ref_shape[0]), int(ref_shape[1]))
    D = depth_values.shape[1]
    with torch.no_grad():   # module.py:127: the sampling grid carries no gradient
        u, v, _ = sweep_pixel_coords(src_proj, ref_proj, depth_values, (h, w))
        gx = u / ((ws - 1) / 2) - 1
        gy = v / ((hs - 1) / 2) - 1
        grid = torch.stack((gx, gy), dim=3).clamp(-10, 10)  # [B,D,h*w,2]
    out = F.grid_sample(src_fea, grid.view(B, D * h, w, 2), mode="bilinear", padding_mode="zeros", align_corners=True)
    return out.view(B, C, D, h, w)


# --------------------------------------------------------------------------
# A4 / A4s: cost aggregation
# --------------------------------------------------------------------------
def variance_cost(ref_fea, warped: Sequence[torch.Tensor]) -> torch.Tensor:
    """``sum f^2 / N - (sum f)^2 / N^2`` over the reference (broadcast over D) and
    the warped sources.  Reference ``models/MVSNet/model.py:113-139``."""
    D = warped[0].shape[2]
    N = len(warped) + 1
    ref_vol = ref_fea.unsqueeze(2).expand(-1, -1, D, -1, -1)
    s = ref_vol.clone()
    sq = ref_vol ** 2
    for wv in warped:
        s = s + wv
        sq = sq + wv ** 2
    return sq / N - (s ** 2) / (N ** 2)


def softmin_cost(ref_fea, warped: Sequence[torch.Tensor], temp: torch.Tensor) -> torch.Tensor:
    """MVSNet-s aggregation, reference ``models/MVSNet/model.py:141-173``:
    per source ``diff = (ref - warp)^2``, ``e = exp(-temp * sum_c diff)``,
    ``cost = sum_v e diff / (sum_v e + 1e-6)``."""
    ref_vol = ref_fea.unsqueeze(2)
    sum_e = 0.0
    sum_v = 0.0
    for wv in warped:
        diff = (ref_vol - wv) ** 2
        e = torch.exp(-temp * diff.sum(dim=1, keepdim=True))
        sum_e = sum_e + e
        sum_v = sum_v + e * diff
    return sum_v / (sum_e + 1e-6)


def variance_cost_streaming(ref_fea, src_feas, ref_proj, src_projs, depth_values) -> torch.Tensor:
    """The same statistic in the reference's EVAL-mode order (``models/MVSNet/model.py:118-137`` with
    ``self.training == False``): one source view at a time, the warped volume folded into the two running sums in
    place and dropped, the final ``sq / N - s^2 / N^2`` in place as well.  Identical values to ``variance_cost`` (same
    operations in the same order per element); it exists because it is what the reference's CPU inference path costs:
    one 503 MB warped volume alive at a time instead of V - 1 of them plus out-of-place sums (bench.py's
    ``cpu_baseline`` leg times this form)."""
    D = depth_values.shape[1]
    N = len(src_feas) + 1
    s = ref_fea.unsqueeze(2).repeat(1, 1, D, 1, 1)
    sq = s ** 2
    for sf, sp in zip(src_feas, src_projs):
        wv = homo_warping(sf, sp, ref_proj, depth_values, ref_fea.shape[-2:])
        s += wv
        sq += wv.pow_(2)
        del wv
    return sq.div_(N).sub_(s.pow_(2).div_(N ** 2))


def build_cost_volume(ref_fea, src_feas, ref_proj, src_projs, depth_values, aggregation="variance", temp=None,
                      streaming: bool = False):
    if aggregation == "variance" and streaming:
        return variance_cost_streaming(ref_fea, src_feas, ref_proj, src_projs, depth_values)
    warped = [homo_warping(sf, sp, ref_proj, depth_values, ref_fea.shape[-2:]) for sf, sp in zip(src_feas, src_projs)]
    if aggregation == "variance":
        return variance_cost(ref_fea, warped)
    if aggregation == "softmin":
        return softmin_cost(ref_fea, warped, temp)
    raise NotImplementedError(aggregation)


# --------------------------------------------------------------------------
# conv blocks (functional)
# --------------------------------------------------------------------------
BN_MOMENTUM = 0.1  # nn.BatchNorm default


class _StoreFn(torch.autograd.Function):
    """Emulates a tensor that the engine keeps in 16-bit storage: the value is rounded on the way forward and its
    gradient on the way back (the engine stores both in that format); arithmetic stays fp32."""

    @staticmethod
    def forward(ctx, x, dtype, round_grad):
        ctx.dtype, ctx.round_grad = dtype, round_grad
        return x.to(dtype).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return (g.to(ctx.dtype).to(g.dtype) if ctx.round_grad else g), None, None


class _GradStoreFn(torch.autograd.Function):
    """Value untouched (fp32 logits), gradient rounded to the 16-bit storage format."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.dtype = dtype
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dtype).to(g.dtype), None


def stored(x: torch.Tensor, dtype: Optional[torch.dtype], round_grad: bool = True) -> torch.Tensor:
    """Identity when ``dtype`` is None (the fp32 reference semantics); otherwise the 16-bit storage emulation used by
    the tests to separate storage-precision effects from implementation errors."""
    return x if dtype is None else _StoreFn.apply(x, dtype, round_grad)


def _bn(x, sd: SD, prefix: str, training: bool = False, new_stats: Optional[dict] = None):
    """eval: running statistics.  training: batch statistics (``nn.BatchNorm*.train()``); the running statistics
    the module would hold after the step are written to ``new_stats`` (the state dict itself is not modified)."""
    if not training:
        return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                            sd[prefix + ".weight"], sd[prefix + ".bias"], training=False, eps=BN_EPS)
    rm, rv = sd[prefix + ".running_mean"].detach().clone(), sd[prefix + ".running_var"].detach().clone()
    y = F.batch_norm(x, rm, rv, sd[prefix + ".weight"], sd[prefix + ".bias"], training=True, momentum=BN_MOMENTUM, eps=BN_EPS)
    if new_stats is not None:
        new_stats[prefix + ".running_mean"], new_stats[prefix + ".running_var"] = rm, rv
    return y


def _w(sd: SD, key: str, store):
    """Conv weight as the MFMA kernels see it under storage emulation (16-bit operands; the gradient stays fp32)."""
    return stored(sd[key], store, round_grad=False)


def conv_bn_relu_3d(x, sd: SD, prefix: str, stride: int = 1, training: bool = False, new_stats: Optional[dict] = None,
                    store=None, skip=None):
    """``ConvBnReLU3D`` reference ``models/MVSNet/module.py:41-48``.  (``store``: 16-bit storage emulation of the raw conv
    output and of the block output; ``skip`` is added after the ReLU, before the block output is stored.)"""
    y = stored(F.conv3d(x, _w(sd, prefix + ".conv.weight", store), None, stride=stride, padding=1), store)
    out = F.relu(_bn(y, sd, prefix + ".bn", training, new_stats))
    return stored(out if skip is None else skip + out, store)


def deconv_bn_relu_3d(x, sd: SD, prefix: str, stride: int = 2, output_padding: int = 1, training: bool = False,
                      new_stats: Optional[dict] = None, store=None, skip=None):
    """``Sequential(ConvTranspose3d(k3,p1), BatchNorm3d, ReLU)`` reference ``model.py:57-70``."""
    y = F.conv_transpose3d(x, _w(sd, prefix + ".0.weight", store), None, stride=stride, padding=1, output_padding=output_padding)
    out = F.relu(_bn(stored(y, store), sd, prefix + ".1", training, new_stats))
    return stored(out if skip is None else skip + out, store)


def cost_reg_net(cost: torch.Tensor, sd: SD, prefix: str = "cost_regularization", taps: Optional[dict] = None,
                 training: bool = False, new_stats: Optional[dict] = None, store=None):
    """MVSNet ``CostRegNet.forward`` reference ``models/MVSNet/model.py:74-84``.
    [B,32,D,h,w] -> [B,1,D,h,w].  ``taps`` (optional dict) receives every layer output."""
    p = prefix + "."
    kw = dict(training=training, new_stats=new_stats, store=store)
    c0 = conv_bn_relu_3d(cost, sd, p + "conv0", **kw)
    c1 = conv_bn_relu_3d(c0, sd, p + "conv1", stride=2, **kw)
    c2 = conv_bn_relu_3d(c1, sd, p + "conv2", **kw)
    c3 = conv_bn_relu_3d(c2, sd, p + "conv3", stride=2, **kw)
    c4 = conv_bn_relu_3d(c3, sd, p + "conv4", **kw)
    c5 = conv_bn_relu_3d(c4, sd, p + "conv5", stride=2, **kw)
    c6 = conv_bn_relu_3d(c5, sd, p + "conv6", **kw)
    u7 = deconv_bn_relu_3d(c6, sd, p + "conv7", skip=c4, **kw)
    u9 = deconv_bn_relu_3d(u7, sd, p + "conv9", skip=c2, **kw)
    u11 = deconv_bn_relu_3d(u9, sd, p + "conv11", skip=c0, **kw)
    logits = F.conv3d(u11, _w(sd, p + "prob.weight", store), sd[p + "prob.bias"], stride=1, padding=1)
    if store is not None:   # the engine hands d loss / d logits to the conv kernels as a 16-bit volume
        logits = _GradStoreFn.apply(logits, store)
    if taps is not None:
        taps.update(conv0=c0, conv1=c1, conv2=c2, conv3=c3, conv4=c4, conv5=c5, conv6=c6, up7=u7, up9=u9, up11=u11,
                    logits=logits)
    return logits


def conv_bn_relu_2d(x, sd: SD, prefix: str, stride: int, pad: int, training: bool = False, store=None):
    y = F.relu(_bn(F.conv2d(x, _w(sd, prefix + ".conv.weight", store), None, stride=stride, padding=pad), sd, prefix + ".bn", training))
    return stored(y, store, round_grad=False)


def feature_net(img: torch.Tensor, sd: SD, prefix: str = "feature", training: bool = False, store=None) -> torch.Tensor:
    """2-D ``FeatureNet`` reference ``models/MVSNet/model.py:21-41`` (upstream of the hot
    path; restated so that the full ``forward()`` can be checked).  [B,3,H,W] -> [B,32,H/4,W/4].
    ``store``: emulate the engine's 2-D extractor, which keeps every layer's output (and the weights) in 16-bit storage."""
    p = prefix + "."
    x = conv_bn_relu_2d(img, sd, p + "conv0", 1, 1, training, store)
    x = conv_bn_relu_2d(x, sd, p + "conv1", 1, 1, training, store)
    x = conv_bn_relu_2d(x, sd, p + "conv2", 2, 2, training, store)
    x = conv_bn_relu_2d(x, sd, p + "conv3", 1, 1, training, store)
    x = conv_bn_relu_2d(x, sd, p + "conv4", 1, 1, training, store)
    x = conv_bn_relu_2d(x, sd, p + "conv5", 2, 2, training, store)
    x = conv_bn_relu_2d(x, sd, p + "conv6", 1, 1, training, store)
    return F.conv2d(x, _w(sd, p + "feature.weight", store), sd[p + "feature.bias"], stride=1, padding=1)


# --------------------------------------------------------------------------
# A6 / A6c: regression and confidence
# --------------------------------------------------------------------------
def depth_regression(prob: torch.Tensor, depth_values: torch.Tensor) -> torch.Tensor:
    """``sum_d p d``; planes per batch [B,D] or per pixel [B,D,h,w].
    Reference ``models/MVSNet/module.py:174-178``."""
    if depth_values.dim() <= 3:
        depth_values = depth_values.view(*depth_values.shape, 1, 1)
    return torch.sum(prob * depth_values, 1)


def photometric_confidence(prob: torch.Tensor) -> torch.Tensor:
    """Sum of the probabilities of planes i-1..i+2 around ``i = trunc(E[index])``
    (zero padded 1 before / 2 after).  Reference ``models/MVSNet/model.py:211-215``."""
    D = prob.shape[1]
    padded = F.pad(prob.unsqueeze(1), pad=(0, 0, 0, 0, 1, 2))
    sum4 = 4 * F.avg_pool3d(padded, (4, 1, 1), stride=1, padding=0).squeeze(1)
    idx = depth_regression(prob, torch.arange(D, dtype=torch.float)).long()
    return torch.gather(sum4, 1, idx.unsqueeze(1)).squeeze(1)


def regress(logits: torch.Tensor, depth_values: torch.Tensor):
    """softmax over D (no sign flip, reference ``model.py:207-209``) + depth + confidence.
    logits [B,D,h,w]."""
    prob = F.softmax(logits, dim=1)
    return prob, depth_regression(prob, depth_values), photometric_confidence(prob)


# --------------------------------------------------------------------------
# whole path
# --------------------------------------------------------------------------
def hot_path(features: Sequence[torch.Tensor], proj: torch.Tensor, depth_values: torch.Tensor, sd: SD,
             aggregation: str = "variance", reference_frame: int = 0, taps: Optional[dict] = None, training: bool = False,
             new_stats: Optional[dict] = None, store=None, streaming: bool = False):
    """Features + cameras -> depth, confidence (the timed region of bench.py).  ``streaming``: build the variance volume
    in the reference's eval-mode order (view by view, in place; same values).

    ``features``: V tensors [B,32,h,w]; ``proj`` [B,V,4,4]; ``depth_values`` [B,V,D].
    Reference ``models/MVSNet/model.py:197-215``."""
    V = len(features)
    ref_fea = features[reference_frame]
    src_feas = [features[i] for i in range(V) if i != reference_frame]
    ref_proj = proj[:, reference_frame]
    src_projs = [proj[:, i] for i in range(V) if i != reference_frame]
    dv = depth_values[:, reference_frame]
    cost = build_cost_volume(ref_fea, src_feas, ref_proj, src_projs, dv, aggregation, sd.get("temp"), streaming=streaming)
    cost = stored(cost, store)
    logits = cost_reg_net(cost, sd, taps=taps, training=training, new_stats=new_stats, store=store).squeeze(1)
    prob, depth, conf = regress(logits, dv)
    if taps is not None:
        taps.update(cost_volume=cost, prob=prob)
    return depth, conf


def forward(imgs, K, R, t, depth_min, depth_max, sd: SD, num_depth: int = 192, aggregation: str = "variance",
            reference_frame: int = 0, taps: Optional[dict] = None, training: bool = False,
            new_stats: Optional[dict] = None, store=None, store_feature_layers: bool = False) -> Dict[str, object]:
    """Full ``MVSNet.forward`` reference ``models/MVSNet/model.py:178-218``.  ``store`` (a 16-bit dtype) emulates the engine's
    storage: every tensor it keeps in HBM is rounded once, arithmetic stays fp32; ``store_feature_layers`` extends that to the
    eight layers of the 2-D extractor (the engine's own extractor; the PyTorch-ROCm one rounds the final map only)."""
    if isinstance(imgs, torch.Tensor):
        imgs = list(torch.unbind(imgs, 1))
    proj, depth_values = mvsnet_cameras(K, R, t, depth_min, depth_max, num_depth)
    feats = [stored(feature_net(im, sd, training=training, store=store if store_feature_layers else None), store, round_grad=False)
             for im in imgs]   # per view, model.py:101-107
    if taps is not None:
        taps.update(features=feats, proj=proj, depth_values=depth_values)
    depth, conf = hot_path(feats, proj, depth_values, sd, aggregation, reference_frame, taps, training, new_stats, store)
    return {"depth": depth, "depth_est_list": [depth], "depth_pair_list": [], "photometric_confidence": conf}
