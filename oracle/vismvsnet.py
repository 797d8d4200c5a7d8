"""CPU restatement of the Vis-MVSNet hot path (3-stage cascade, pair-wise group correlation, visibility-weighted
fusion).  TEST INFRASTRUCTURE ONLY.

Functional form driven by the reference's state dict (keys of ``models.VisMVSNet.frontend.Frontend``).  fp32,
eval-mode BatchNorm.  Every function cites the reference file:line it follows.
"""
from __future__ import annotations

from typing import Dict, List, Mapping, Optional, Sequence

import torch
import torch.nn.functional as F

SD = Mapping[str, torch.Tensor]
BN_EPS = 1e-5


# train() mode of the modules (the reference under train.py): BatchNorm uses batch statistics; the running statistics the
# modules would hold afterwards are collected in MODE["new_stats"] (chained across repeated calls of the same module: the
# pair U-Net runs once per source view).  Set through the ``train_mode`` context manager; gradients then come from ATen
# autograd through these same functions, exactly as in the reference.
MODE = {"training": False, "new_stats": None, "store": None}


class storage:
    """Emulates the engine's 16-bit HBM storage inside this oracle (tests only): every tensor the engine keeps in 16-bit form
    -- feature maps, conv weights, the correlation volume, every U-Net layer output, the fused volume -- is rounded once to
    ``dtype``; arithmetic stays fp32 and the 1-channel score heads stay fp32 (as in the engine).  The result is what an ideal
    pipeline with that storage format computes: the yardstick for the engine's bf16 / fp16 parity bars."""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        self.prev = MODE["store"]
        MODE["store"] = self.dtype
        return self

    def __exit__(self, *exc):
        MODE["store"] = self.prev
        return False


def _q(x):
    return x if MODE["store"] is None else x.to(MODE["store"]).to(x.dtype)


class train_mode:
    def __init__(self, new_stats: Optional[dict] = None):
        self.new_stats = new_stats

    def __enter__(self):
        self.prev = dict(MODE)
        MODE.update(training=True, new_stats=self.new_stats)
        return self

    def __exit__(self, *exc):
        MODE.update(self.prev)
        return False


def _bn(x, sd: SD, p: str):
    if not MODE["training"]:
        return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                            training=False, eps=BN_EPS)
    ns = MODE["new_stats"]
    src = ns if (ns is not None and p + ".running_mean" in ns) else sd
    rm, rv = src[p + ".running_mean"].detach().clone(), src[p + ".running_var"].detach().clone()
    y = F.batch_norm(x, rm, rv, sd[p + ".weight"], sd[p + ".bias"], training=True, momentum=0.1, eps=BN_EPS)
    if ns is not None:
        ns[p + ".running_mean"], ns[p + ".running_var"] = rm, rv
    return y


def _conv(x, w, stride=1, dim=2, padding=None):
    pad = (w.shape[-1] // 2) if padding is None else padding
    return (F.conv2d if dim == 2 else F.conv3d)(x, _q(w), None, stride=stride, padding=pad)


def _deconv(x, w, dim=2):
    return (F.conv_transpose2d if dim == 2 else F.conv_transpose3d)(x, _q(w), None, stride=2, padding=1, output_padding=1)


# --------------------------------------------------------------------------
# generic blocks: models/VisMVSNet/nn_utils.py:123-278
# --------------------------------------------------------------------------
def basic_block(x, sd: SD, p: str, stride: int, dim: int):
    """``BasicBlock`` nn_utils.py:123-171: conv-bn-relu-conv-bn (+ 1x1 strided conv-bn shortcut when present), relu."""
    out = _q(F.relu(_bn(_conv(x, sd[p + ".conv1.weight"], stride, dim), sd, p + ".bn1")))
    out = _bn(_conv(out, sd[p + ".conv2.weight"], 1, dim), sd, p + ".bn2")
    if (p + ".downsample.0.weight") in sd:
        x = _q(_bn(_conv(x, sd[p + ".downsample.0.weight"], stride, dim, padding=0), sd, p + ".downsample.1"))
    return _q(F.relu(out + x))


def layer(x, sd: SD, p: str, blocks: int, stride: int, dim: int):
    """``_make_layer`` nn_utils.py:174-191: first block carries the stride, the rest are stride 1."""
    for i in range(blocks):
        x = basic_block(x, sd, f"{p}.{i}", stride if i == 0 else 1, dim)
    return x


def unet(x, sd: SD, p: str, enc_names: Sequence[str], dec_names: Sequence[str], enc_blocks: int, dec_blocks: int,
         dim: int, multi_scale: int = 1):
    """``UNet.forward`` nn_utils.py:258-278 for the two configurations the model uses (no bottom / head blocks):
    encoder layers (first stride 1, then stride 2), decoder = deconv -> cat([deconv, skip]) -> conv (-> res block)."""
    enc_out = []
    for i, name in enumerate(enc_names):
        x = layer(x, sd, f"{p}.enc_blocks.{name}", enc_blocks, 1 if i == 0 else 2, dim)
        enc_out.append(x)
    dec_out = [x]
    for i, name in enumerate(dec_names):
        q = f"{p}.dec_blocks.{name}"
        x = _q(_deconv(x, sd[q + ".0.weight"], dim))
        x = torch.cat([x, enc_out[-2 - i]], 1)            # deconv channels first, nn_utils.py:269-271
        x = _q(_conv(x, sd[q + ".1.weight"], 1, dim))
        if dec_blocks > 0:
            x = layer(x, sd, q + ".2", dec_blocks, 1, dim)
        dec_out.append(x)
    return x if multi_scale == 1 else dec_out[-multi_scale:]


def feat_ext(img, sd: SD, p: str = "model.feat_ext"):
    """``FeatExt`` model_cas.py:18-35: three 32-channel maps at 1/8, 1/4, 1/2 resolution (upstream of the hot path)."""
    x = _q(F.relu(_bn(_conv(img, sd[p + ".init_conv.0.weight"], 2, 2), sd, p + ".init_conv.1")))
    o1, o2, o3 = unet(x, sd, p + ".unet", ["2d2_0", "2d4_1", "2d8_2"], ["2d16_3", "2d8_4"], 2, 1, 2, multi_scale=3)
    return (_q(_conv(o1, sd[p + ".final_conv_1.weight"])), _q(_conv(o2, sd[p + ".final_conv_2.weight"])),
            _q(_conv(o3, sd[p + ".final_conv_3.weight"])))


def reg_unet(x, sd: SD, p: str, tag: str):
    """3-D ``UNet(8, 1, 0, 4, [], [8, 16], [], tag, dim=3)`` model_cas.py:43,67: [n,8,d,h,w] -> [n,8,d,h,w]."""
    return unet(x, sd, p + ".unet", [f"{tag}4_0", f"{tag}8_1"], [f"{tag}16_2"], 1, 0, 3)


def uncert_net(x, sd: SD, p: str):
    """``UncertNet`` model_cas.py:77-98: [n,1,h,w] entropy map -> [n,1,h,w] log-uncertainty (one head)."""
    out = F.relu(_bn(_conv(x, sd[p + ".conv1.0.weight"]), sd, p + ".conv1.1"))
    out = F.relu(_bn(_conv(out, sd[p + ".conv2.0.weight"]), sd, p + ".conv2.1"))
    out = out + x
    return _conv(out, sd[p + ".head_convs.0.weight"])


# --------------------------------------------------------------------------
# A3: homographies and warp -- models/VisMVSNet/homography.py
# --------------------------------------------------------------------------
def scale_camera(cam, scale: float):
    """preproc.py:63-92: focal lengths and principal point times ``scale``."""
    new = cam.clone()
    new[..., 1, 0, 0] = cam[..., 1, 0, 0] * scale
    new[..., 1, 1, 1] = cam[..., 1, 1, 1] * scale
    new[..., 1, 0, 2] = cam[..., 1, 0, 2] * scale
    new[..., 1, 1, 2] = cam[..., 1, 1, 2] * scale
    return new


def get_homographies(left_cam, right_cam, depth_num: int, depth_start, depth_interval, inv: bool = False):
    """homography.py:23-74: ``H_d = K_r R_r (I - (c_r - c_l) n_l^T / (d + 1e-9)) R_l^T K_l^-1`` per plane (and per
    pixel when depth_start is [n,1,h,w]); ``inv`` = planes uniform in inverse depth between the same end points (:41-46).
    Returns [n, d, 1|h, 1|w, 3, 3]."""
    n = left_cam.shape[0]
    R_l, R_r = left_cam[:, 0, :3, :3], right_cam[:, 0, :3, :3]
    t_l, t_r = left_cam[:, 0, :3, 3:4], right_cam[:, 0, :3, 3:4]
    K_l, K_r = left_cam[:, 1, :3, :3], right_cam[:, 1, :3, :3]
    steps = torch.arange(depth_num, dtype=left_cam.dtype).view(1, depth_num, 1, 1)
    if not inv:
        depth = depth_start + depth_interval * steps
    else:                                                           # homography.py:41-46
        depth_end = depth_start + (depth_num - 1) * depth_interval
        inv_interv = (1 / (depth_start + 1e-9) - 1 / (depth_end + 1e-9)) / (depth_num - 1 + 1e-9)
        depth = 1 / (1 / (depth_end + 1e-9) + inv_interv * steps)
    depth = depth.unsqueeze(-1).unsqueeze(-1)                       # [n,d,1|h,1|w,1,1]
    K_l_inv = K_l.float().inverse()
    fronto = R_l[:, 2:3, :3]                                        # [n,1,3]
    c_l = -R_l.transpose(-2, -1) @ t_l
    c_r = -R_r.transpose(-2, -1) @ t_r
    temp = ((c_r - c_l) @ fronto).view(n, 1, 1, 1, 3, 3)
    mid0 = torch.eye(3).view(1, 1, 1, 1, 3, 3) - temp / (depth + 1e-9)
    mid1 = (R_l.transpose(-2, -1) @ K_l_inv).view(n, 1, 1, 1, 3, 3)
    return K_r.view(n, 1, 1, 1, 3, 3) @ R_r.view(n, 1, 1, 1, 3, 3) @ (mid0 @ mid1)


def homography_warping(src, H, ref_shape):
    """homography.py:77-120: half-pixel centres, ``z <= 0`` -> (-10,-10), coordinates / size * 2 - 1 clamped to
    +-1.1, ``grid_sample(align_corners=True)``.  src [m,c,hs,ws], H [m,1|h,1|w,3,3] -> [m,c,h,w]."""
    h, w = ref_shape
    with torch.no_grad():   # homography.py:92,110
        xs = (torch.arange(w, dtype=torch.float32) + 0.5).repeat(h, 1)
        ys = (torch.arange(h, dtype=torch.float32) + 0.5).repeat(w, 1).t()
        grid = torch.stack([xs, ys, torch.ones_like(xs)], dim=-1).unsqueeze(-1).unsqueeze(0)   # [1,h,w,3,1]
        hom = (H @ grid).squeeze(-1)                                                            # [m,h,w,3]
        valid = hom[..., 2] > 0
        coord = hom[..., :2] / hom[..., 2:3].clamp(min=1e-9)
        coord = torch.where(valid.unsqueeze(-1), coord, torch.full_like(coord, -10.0))
        norm = coord.clone()
        norm[..., 0] /= src.shape[3]
        norm[..., 1] /= src.shape[2]
        norm = (norm * 2 - 1).clamp(-1.1, 1.1)
    return F.grid_sample(src, norm, mode="bilinear", padding_mode="zeros", align_corners=True)


def warp_volume(src_feat, ref_cam, src_cam, depth_num, depth_start, depth_interval, s_scale, ref_shape):
    """``SingleStage.build_cost_volume`` model_cas.py:176-186 -> warped source volume [n,c,d,h,w]."""
    with torch.no_grad():   # homography.py:25: the homographies carry no gradient
        rc, sc = scale_camera(ref_cam, 1.0 / s_scale), scale_camera(src_cam, 1.0 / s_scale)
        Hs = get_homographies(rc, sc, depth_num, depth_start, depth_interval)        # [n,d,1|h,1|w,3,3]
    n, c = src_feat.shape[:2]
    src_rep = src_feat.unsqueeze(1).repeat(1, depth_num, 1, 1, 1).view(-1, *src_feat.shape[1:])
    warped = homography_warping(src_rep, Hs.reshape(-1, *Hs.shape[2:]), ref_shape)
    return warped.view(n, depth_num, c, *ref_shape).transpose(1, 2)


# --------------------------------------------------------------------------
# A4g / A6v
# --------------------------------------------------------------------------
def groupwise_correlation(v1, v2, groups: int = 8):
    """nn_utils.py:473-490: sum (not mean) of the products inside each channel group. [n,c,...] -> [n,groups,...]."""
    n, c = v1.shape[:2]
    return (v1.reshape(n, groups, c // groups, *v1.shape[2:]) * v2.reshape(n, groups, c // groups, *v2.shape[2:])).sum(2)


def soft_argmin(volume, window: Optional[float] = None):
    """nn_utils.py:453-466 with dim=1, keepdim=True: softmax over planes, expected index, optional window prob."""
    prob = F.softmax(volume, dim=1)
    idx = torch.arange(volume.shape[1], dtype=prob.dtype).view(1, -1, 1, 1)
    out = torch.sum(idx * prob, dim=1, keepdim=True)
    if window is None:
        return prob, out
    mask = ((idx - out).abs() <= window).to(volume.dtype)
    return prob, out, torch.sum(prob * mask, dim=1, keepdim=True)


def entropy(prob):
    """nn_utils.py:469-470."""
    return torch.sum(-prob * prob.clamp(1e-9, 1.0).log(), dim=1, keepdim=True)


# --------------------------------------------------------------------------
# one cascade stage: model_cas.py:303-420 (mode 'soft')
# --------------------------------------------------------------------------
def single_stage(ref_feat, ref_cam, srcs_feat, srcs_cam, sd: SD, p: str, depth_num: int, depth_start, depth_interval,
                 s_scale: int, taps: Optional[dict] = None):
    n, c, h, w = ref_feat.shape
    ref_vol = ref_feat.unsqueeze(2).repeat(1, 1, depth_num, 1, 1)
    fused = torch.zeros(n, 8, depth_num, h, w)
    weight_sum = torch.zeros(n, 1, 1, h, w)
    pair_results = []
    for i, (sf, sc) in enumerate(zip(srcs_feat, srcs_cam)):
        warped = warp_volume(sf, ref_cam, sc, depth_num, depth_start, depth_interval, s_scale, (h, w))
        cost = _q(groupwise_correlation(ref_vol, warped, 8))
        interm = reg_unet(cost, sd, p + ".reg", "reg1")
        score = _conv(interm, sd[p + ".reg_pair.final_conv.weight"], 1, 3).squeeze(1)
        prob, est_class = soft_argmin(score)
        est_depth = est_class * depth_interval + depth_start
        ent = entropy(prob)
        uncert = uncert_net(ent, sd, p + ".uncert_net")
        pair_results.append([est_depth, [uncert]])
        weight = (-uncert).exp().unsqueeze(2)
        weight_sum = weight_sum + weight
        fused = fused + interm * weight
        if taps is not None and i == 0:
            taps.update(warped0=warped, cost0=cost, interm0=interm, score0=score, entropy0=ent, uncert0=uncert)
    fused = _q(fused / weight_sum)
    score = _conv(reg_unet(fused, sd, p + ".reg_fuse", "reg2"), sd[p + ".reg_fuse.final_conv.weight"], 1, 3).squeeze(1)
    prob, est_class, prob_map = soft_argmin(score, window=2)
    est_depth = est_class * depth_interval + depth_start
    if taps is not None:
        taps.update(fused=fused, score=score)
    return est_depth, prob_map, pair_results


def fill_cam_array(K, R, t, start_depth, depth_interval):
    """frontend.py:14-24: [b,2,4,4] = ([R|t], [K ; (start, interval)])."""
    b = K.shape[0]
    res = torch.zeros(b, 2, 4, 4)
    res[:, 0, :3, :3] = R
    res[:, 0, :3, 3:4] = t
    res[:, 1, :3, :3] = K
    res[:, 1, 3, 0] = start_depth
    res[:, 1, 3, 1] = depth_interval
    return res


def forward(imgs, K, R, t, depth_min, depth_max, sd: SD, depth_nums=(32, 16, 8), interval_scales=(4, 2, 1),
            attr_interval_scales=(4, 2, 1), reference_frame: int = 0, taps: Optional[dict] = None):
    """``Frontend.forward`` frontend.py:26-109.  ``interval_scales`` is the (optional) kwarg, ``attr_interval_scales``
    the instance attribute: the stage-2/3 start offsets use the ATTRIBUTE (frontend.py:76-78,89-91), as the reference does."""
    depth_interval = (depth_max - depth_min) / 128
    if isinstance(imgs, torch.Tensor):
        imgs = list(torch.unbind(imgs, 1))
    v = len(imgs)
    src_idx = [i for i in range(v) if i != reference_frame]
    ref, src = imgs[reference_frame], [imgs[i] for i in src_idx]
    n = ref.shape[0]
    ref_cam = fill_cam_array(K[:, reference_frame], R[:, reference_frame], t[:, reference_frame],
                             depth_min[:, reference_frame], depth_interval[:, reference_frame])
    srcs_cam = [fill_cam_array(K[:, i], R[:, i], t[:, i], depth_min[:, i], depth_interval[:, i]) for i in src_idx]
    rf = feat_ext(ref, sd)
    sf = [feat_ext(s, sd) for s in src]
    di = depth_interval[:, reference_frame].view(n, 1, 1, 1)
    ds1 = ref_cam[:, 1:2, 3:4, 0:1]
    stage_taps = [dict() if taps is not None else None for _ in range(3)]
    d1, p1, pr1 = single_stage(rf[0], ref_cam, [f[0] for f in sf], srcs_cam, sd, "model.stage1", depth_nums[0], ds1,
                               di * interval_scales[0], 8, stage_taps[0])
    p1_up = F.interpolate(p1, scale_factor=4, mode="bilinear", align_corners=False)
    ds2 = F.interpolate(d1.detach(), size=rf[1].shape[2:], mode="bilinear", align_corners=False) \
        - depth_nums[1] * di * attr_interval_scales[1] / 2
    d2, p2, pr2 = single_stage(rf[1], ref_cam, [f[1] for f in sf], srcs_cam, sd, "model.stage2", depth_nums[1], ds2,
                               di * interval_scales[1], 4, stage_taps[1])
    p2_up = F.interpolate(p2, scale_factor=2, mode="bilinear", align_corners=False)
    ds3 = F.interpolate(d2.detach(), size=rf[2].shape[2:], mode="bilinear", align_corners=False) \
        - depth_nums[2] * di * attr_interval_scales[2] / 2
    d3, p3, pr3 = single_stage(rf[2], ref_cam, [f[2] for f in sf], srcs_cam, sd, "model.stage3", depth_nums[2], ds3,
                               di * interval_scales[2], 2, stage_taps[2])
    if taps is not None:
        taps.update(features_ref=rf, features_src=sf, ref_cam=ref_cam, srcs_cam=srcs_cam, stages=stage_taps,
                    depth_starts=[ds1, ds2, ds3])
    return {"depth": d3.squeeze(1), "depth_est_list": [d3.squeeze(1), d2.squeeze(1), d1.squeeze(1)],
            "depth_pair_list": [pr3, pr2, pr1], "photometric_confidence": torch.cat([p1_up, p2_up, p3], dim=1)}
