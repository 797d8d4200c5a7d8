"""CPU restatement of the CVP-MVSNet hot path (coarse-to-fine cost-volume pyramid).  TEST INFRASTRUCTURE ONLY.

Functional form driven by the reference's state dict (keys of ``models.CVP_MVSNet.frontend.Frontend``); fp32 except
where the reference itself switches to fp64 (``calDepthHypo``).  Every function cites the reference file:line.
"""
from __future__ import annotations

from typing import Dict, List, Mapping, Optional, Sequence

import torch
import torch.nn.functional as F

SD = Mapping[str, torch.Tensor]
BN_EPS = 1e-5
PYR = ("conv0aa", "conv0ba", "conv0bb", "conv0bc", "conv0bd", "conv0be", "conv0bf", "conv0bg", "conv0bh")
STORE = {"dtype": None}


class storage:
    """Emulates the engine's 16-bit HBM storage inside this oracle (tests only): pyramid layers, conv weights, cost volumes
    and every U-Net layer output are rounded once to ``dtype``; arithmetic and the 1-channel logits stay fp32.  The yardstick
    for the engine's bf16 / fp16 parity bars (what an ideal pipeline with that storage format computes)."""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        self.prev = STORE["dtype"]
        STORE["dtype"] = self.dtype
        return self

    def __exit__(self, *exc):
        STORE["dtype"] = self.prev
        return False


def _q(x):
    return x if STORE["dtype"] is None else x.to(STORE["dtype"]).to(x.dtype)


def _bn(x, sd: SD, p: str, training: bool = False, new_stats: Optional[dict] = None):
    """eval: running statistics; training: batch statistics of ``nn.BatchNorm3d.train()`` -- the running statistics the
    module would hold afterwards go to ``new_stats`` (read from there first, so several passes through the same net chain
    their updates like the reference's module does when it is called once per pyramid level)."""
    if not training:
        return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                            training=False, eps=BN_EPS)
    src = new_stats if (new_stats is not None and p + ".running_mean" in new_stats) else sd
    rm, rv = src[p + ".running_mean"].detach().clone(), src[p + ".running_var"].detach().clone()
    y = F.batch_norm(x, rm, rv, sd[p + ".weight"], sd[p + ".bias"], training=True, momentum=0.1, eps=BN_EPS)
    if new_stats is not None:
        new_stats[p + ".running_mean"], new_stats[p + ".running_var"] = rm, rv
    return y


def feature_pyramid(img, sd: SD, scales: int, p: str = "model.featurePyramid"):
    """``FeaturePyramid`` net.py:21-47: nine conv + LeakyReLU(0.1) layers on the image and its bilinear half-size
    copies; 16 channels at EVERY level, finest first (upstream of the hot path)."""
    def tower(x):
        for name in PYR:
            x = _q(F.leaky_relu(F.conv2d(x, _q(sd[f"{p}.{name}.0.weight"]), sd[f"{p}.{name}.0.bias"], padding=1), 0.1))
        return x
    out = [tower(img)]
    for _ in range(scales - 1):
        img = F.interpolate(img, scale_factor=0.5, mode="bilinear", align_corners=None)
        out.append(tower(img))
    return out


def condition_intrinsics(K, img_shape, fp_shapes):
    """modules.py:31-50: rows 0-1 of K divided by image_height / feature_height per level. [B,3,3] -> [B,L,3,3]."""
    outs = []
    for s in fp_shapes:
        r = img_shape[2] / s[2]
        k = K.clone()
        k[:, :2, :] = k[:, :2, :] / r
        outs.append(k)
    return torch.stack(outs).permute(1, 0, 2, 3)


def sweeping_depth_hypos(depth_min, depth_max, n: int):
    """modules.py:53-71: ``d_i = min + i (max - min) / n`` for i < n (divisor n, not n-1).  [B] -> [B,n]."""
    step = (depth_max - depth_min) / n
    return depth_min.unsqueeze(1) + torch.arange(n) * step.unsqueeze(1)


def _proj(K, E):
    last = torch.tensor([[[0.0, 0.0, 0.0, 1.0]]]).repeat(len(K), 1, 1)
    return torch.cat((torch.matmul(K, E[:, 0:3, :]), last), 1)


def homo_warping(src_feature, ref_in, src_in, ref_ex, src_ex, depth_hypos, ref_shape=None):
    """modules.py:74-128 (and the inline copy in proj_cost :241-281): same geometry as MVSNet's warp; depth_hypos is
    [B,D] or per pixel [B,D,h*w] / [B,D,h,w]."""
    B, C, hs, ws = src_feature.shape
    h, w = (hs, ws) if ref_shape is None else (int(ref_shape[0]), int(ref_shape[1]))
    D = depth_hypos.shape[1]
    with torch.no_grad():   # modules.py:83 / :241: the sampling grid carries no gradient (matters in train(): the refinement
        # hypotheses depend on the coarse depth, and the reference lets that reach the depth only through the regression)
        proj = torch.matmul(_proj(src_in, src_ex), torch.inverse(_proj(ref_in, ref_ex)))
        rot, trans = proj[:, :3, :3], proj[:, :3, 3:4]
        yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
        pix = torch.stack((xx.reshape(-1), yy.reshape(-1), torch.ones(h * w))).unsqueeze(0).repeat(B, 1, 1)
        q = torch.matmul(rot, pix).unsqueeze(2) * depth_hypos.reshape(B, 1, D, -1) + trans.view(B, 3, 1, 1)
        uv = q[:, :2] / q[:, 2:3]
        uv = torch.where((q[:, 2:3] <= 0).expand(-1, 2, -1, -1), torch.full_like(uv, -10.0), uv)
        gx = uv[:, 0] / ((ws - 1) / 2) - 1
        gy = uv[:, 1] / ((hs - 1) / 2) - 1
        grid = torch.stack((gx, gy), dim=3).clamp(-10, 10)
    out = F.grid_sample(src_feature, grid.view(B, D * h, w, 2), mode="bilinear", padding_mode="zeros", align_corners=True)
    return out.view(B, C, D, h, w)


def variance_cost(ref_fea, warped: Sequence[torch.Tensor]):
    """net.py:129-152 / modules.py:236-287: ``sum f^2 / N - (sum f / N)^2``."""
    D = warped[0].shape[2]
    N = len(warped) + 1
    s = ref_fea.unsqueeze(2).repeat(1, 1, D, 1, 1)
    sq = s ** 2
    for wv in warped:
        s = s + wv
        sq = sq + wv ** 2
    return _q(sq / N - (s / N) ** 2)


def cbr3(x, sd, p, stride=1, training=False, new_stats=None):
    return _q(F.relu(_bn(F.conv3d(x, _q(sd[p + ".conv.weight"]), None, stride=stride, padding=1), sd, p + ".bn", training, new_stats)))


def cost_reg_net(x, sd: SD, p: str = "model.cost_reg_refine", taps: Optional[dict] = None, training: bool = False,
                 new_stats: Optional[dict] = None):
    """CVP ``CostRegNet`` net.py:50-85: conv0,0a 16->16; conv1 16->32 s2; conv2,2a; conv3 32->64 (stride 1); conv4,4a;
    conv5^T 64->32 (stride 1, op 0) + conv2; conv6^T 32->16 (s2, op 1) + conv0; prob0 16->1.  -> [B,D,h,w]."""
    kw = dict(training=training, new_stats=new_stats)
    c0 = cbr3(cbr3(x, sd, p + ".conv0", **kw), sd, p + ".conv0a", **kw)
    c2 = cbr3(cbr3(cbr3(c0, sd, p + ".conv1", 2, **kw), sd, p + ".conv2", **kw), sd, p + ".conv2a", **kw)
    c4 = cbr3(cbr3(cbr3(c2, sd, p + ".conv3", **kw), sd, p + ".conv4", **kw), sd, p + ".conv4a", **kw)
    c5 = _q(c2 + F.relu(_bn(F.conv_transpose3d(c4, _q(sd[p + ".conv5.0.weight"]), None, stride=1, padding=1, output_padding=0), sd, p + ".conv5.1", **kw)))
    c6 = _q(c0 + F.relu(_bn(F.conv_transpose3d(c5, _q(sd[p + ".conv6.0.weight"]), None, stride=2, padding=1, output_padding=1), sd, p + ".conv6.1", **kw)))
    logits = F.conv3d(c6, _q(sd[p + ".prob0.weight"]), sd[p + ".prob0.bias"], padding=1).squeeze(1)
    if taps is not None:
        taps.update(conv0=c0, conv2=c2, conv4=c4, conv5=c5, conv6=c6, logits=logits)
    return logits


def cal_depth_hypo(ref_depths, ref_in, src_in, ref_ex, src_ex, depth_min, depth_max):
    """Eval-mode hypothesis maps, ``calDepthHypo`` modules.py:131-226: the depth interval that moves the projection in
    the FIRST source view by one pixel along the epipolar line (fp64, per batch item, MEDIAN over valid pixels), then
    8 planes ``depth + k * interval`` for k = -4..3.  ref_depths [B,H,W]; src_in [B,N,3,3]; src_ex [B,N,4,4]."""
    d = 4
    B, H, W = ref_depths.shape
    ri, si = ref_in.double(), src_in.double()
    re, se = ref_ex.double(), src_ex.double()
    hypos = ref_depths.unsqueeze(1).repeat(1, 2 * d, 1, 1)
    for b in range(B):
        xx, yy = torch.meshgrid(torch.arange(W), torch.arange(H), indexing="ij")
        xxx, yyy = xx.reshape(-1).double(), yy.reshape(-1).double()
        X = torch.stack([xxx, yyy, torch.ones_like(xxx)], dim=0)
        D1 = ref_depths[b].transpose(0, 1).reshape(-1)
        D2 = D1 + 1
        ones = torch.ones_like(xxx).unsqueeze(0)
        P1 = se[b][0] @ (torch.inverse(re[b]) @ torch.cat([torch.inverse(ri[b]) @ (X * D1), ones], 0))
        P2 = se[b][0] @ (torch.inverse(re[b]) @ torch.cat([torch.inverse(ri[b]) @ (X * D2), ones], 0))
        X1 = si[b][0] @ P1[:3]
        X1_d = X1[2].clone()
        X1 = X1 / X1_d
        X2 = si[b][0] @ P2[:3]
        X2_d = X2[2].clone()
        X2 = X2 / X2_d
        dirv = X2 - X1
        nrm = torch.norm(dirv, dim=0)
        dirv = dirv / torch.clamp(nrm, min=1e-8)
        X3 = X1 + dirv
        A = (ri[b] @ re[b][:3, :3]) @ torch.inverse(si[b][0] @ se[b][0, :3, :3])
        tmp1 = X1_d * (A @ X1)
        tmp2 = A @ X3
        M1 = torch.cat([X.t().unsqueeze(2), tmp2.t().unsqueeze(2)], 2)[:, 1:, :]
        M2 = tmp1.t()[:, 1:]
        valid = (nrm > 1e-8) & (X1_d > 1e-8) & (X2_d > 1e-8) & (torch.abs(torch.det(M1)) > 1e-8)
        if valid.sum() > 0:
            ans = torch.inverse(M1[valid]) @ M2.unsqueeze(2)[valid]
            delta = ans[:, 0, 0]
        else:
            delta = (depth_max - depth_min) / 128 * torch.ones_like(X1_d)
        interval = torch.abs(delta).median()
        for k in range(-d, d):
            hypos[b, k + d] += k * interval
    return hypos.float()


def forward(imgs, K, R, t, depth_min, depth_max, sd: SD, nscale: int = 2, training_hypos: bool = False,
            reference_frame: int = 0, taps: Optional[dict] = None, training: bool = False, new_stats: Optional[dict] = None):
    """``Frontend.forward`` frontend.py:10-38 + ``network.forward`` net.py:96-229.  ``training_hypos`` selects the
    train()-mode hypothesis rule (48 coarse planes, fixed halving intervals) instead of the eval() one (96 planes,
    calDepthHypo); ``training`` additionally switches BatchNorm to batch statistics (the module in ``train()``: pass both
    for a training step; gradients then come from ATen autograd like in the reference)."""
    if isinstance(imgs, torch.Tensor):
        imgs = list(torch.unbind(imgs, 1))
    V = len(imgs)
    src_idx = [i for i in range(V) if i != reference_frame]
    ref_img, src_imgs = imgs[reference_frame], [imgs[i] for i in src_idx]
    B = ref_img.shape[0]
    bottom = torch.tensor([0.0, 0.0, 0.0, 1.0]).view(1, 1, 4)
    ref_in, src_in = K[:, reference_frame], K[:, src_idx]
    ref_ex = torch.cat((torch.cat((R[:, reference_frame], t[:, reference_frame]), 2), bottom.expand(B, 1, 4)), 1)
    src_ex = torch.cat((torch.cat((R[:, src_idx], t[:, src_idx]), 3), bottom.view(1, 1, 1, 4).expand(B, len(src_idx), 1, 4)), 2)
    dmin, dmax = depth_min[:, reference_frame], depth_max[:, reference_frame]

    ref_pyr = feature_pyramid(ref_img, sd, nscale)
    src_pyrs = [feature_pyramid(s, sd, nscale) for s in src_imgs]
    ref_in_ms = condition_intrinsics(ref_in, ref_img.shape, [f.shape for f in ref_pyr])
    src_in_ms = torch.stack([condition_intrinsics(src_in[:, i], ref_img.shape, [f.shape for f in src_pyrs[i]])
                             for i in range(len(src_idx))]).permute(1, 0, 2, 3, 4)          # [B,N,L,3,3]

    hypos = sweeping_depth_hypos(dmin, dmax, 48 if training_hypos else 96)
    warped = [homo_warping(src_pyrs[i][-1], ref_in_ms[:, -1], src_in_ms[:, i, -1], ref_ex, src_ex[:, i], hypos,
                           ref_pyr[-1].shape[2:]) for i in range(len(src_idx))]
    cost = variance_cost(ref_pyr[-1], warped)
    level_taps = {} if taps is not None else None
    logits = cost_reg_net(cost, sd, taps=level_taps, training=training, new_stats=new_stats)
    prob = F.softmax(logits, dim=1)
    depth = torch.sum(prob * hypos.view(*hypos.shape, 1, 1), 1)
    est = [depth]
    if taps is not None:
        taps.update(coarse_cost=cost, coarse=level_taps, coarse_hypos=hypos, ref_pyr=ref_pyr, src_pyrs=src_pyrs, refine=[])
    for id_level, level in enumerate(range(nscale - 2, -1, -1)):
        depth_up = F.interpolate(depth[None, :], size=None, scale_factor=2, mode="bicubic", align_corners=None).squeeze(0)
        if training_hypos:
            interval = (dmax - dmin) / 48 / 2 ** (id_level + 1)
            hyp = torch.stack([depth_up + i * interval.view(-1, 1, 1) for i in range(-4, 4)], dim=1)
        else:
            hyp = cal_depth_hypo(depth_up, ref_in_ms[:, level], src_in_ms[:, :, level], ref_ex, src_ex, dmin, dmax)
        warped = [homo_warping(src_pyrs[i][level], ref_in_ms[:, level], src_in_ms[:, i, level], ref_ex, src_ex[:, i], hyp,
                               ref_pyr[level].shape[2:]) for i in range(len(src_idx))]
        cost = variance_cost(ref_pyr[level], warped)
        lt = {} if taps is not None else None
        logits = cost_reg_net(cost, sd, taps=lt, training=training, new_stats=new_stats)
        prob = F.softmax(logits, dim=1)
        depth = torch.sum(prob * hyp, 1)
        est.append(depth)
        if taps is not None:
            lt.update(hypos=hyp, cost=cost, depth_up=depth_up)
            taps["refine"].append(lt)
    D = prob.shape[1]
    sum4 = 4 * F.avg_pool3d(F.pad(prob.unsqueeze(1), pad=(0, 0, 0, 0, 1, 2)), (4, 1, 1), stride=1, padding=0).squeeze(1)
    idx = torch.sum(prob * torch.arange(D, dtype=torch.float).view(1, D, 1, 1), 1).long()
    conf = torch.gather(sum4, 1, idx.unsqueeze(1)).squeeze(1)
    est.reverse()
    return {"depth": est[0], "depth_est_list": est, "depth_pair_list": [], "photometric_confidence": conf.unsqueeze(1)}
