"""Parity at the FULL sizes of BASELINE.json's configurations (1), (3), (4), (5) -- the sizes at which other kernel variants
are selected than at the tiny golden sizes (64-channel brick convs, depth-chunk heuristics, 8-source warps, multi-million-voxel
sweeps).  The fp32 oracle cannot run these whole in seconds, so every stage of the engine's full-size run is checked on a
CROPPED WINDOW of reference pixels: the engine's own stage inputs (feature maps, cameras, per-pixel depth hypotheses) are
cropped, the reference camera's principal point is shifted by the window origin (a pinhole crop is exactly that), and the
oracle runs the stage on the window.  Warp + cost are pointwise in the reference pixel, so they are exact on the window; the
3-D U-Nets see the crop border, so their outputs are compared away from it by their receptive radius (on sides that ARE the
image border the zero padding coincides and the comparison goes to the edge).  Config (1) has a golden written by the reference
itself at its real size.  Size-independent properties (shard == unsharded) close the file."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _util import check_close, load_golden, retry_infra, t

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from wild_deep_mvs_amd import _lib as L, ops, synthetic
    return L, ops, synthetic


def _rel_l1(a, b):
    return float((a - b).abs().mean() / b.abs().mean())


# ---------------------------------------------------------------------------------------------------------------------
# configuration (1): MVSNet-s, 1 ref + 2 src, 128x160, D = 48 -- reference golden at the real size
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_config1_mvsnet_s_matches_reference_golden(gpu, dtype):
    L, ops, synthetic = gpu
    from wild_deep_mvs_amd.models.MVSNet.model import MVSNet
    from oracle import mvsnet as O
    g = load_golden("mvsnet_s_cfg1.npz")
    H, W, V, D, seed, scene_seed, gain = [int(x) for x in g["meta"]]
    net = MVSNet("softmin")
    sd = synthetic.sharpened_state_dict("mvsnet", synthetic.template_of(net), seed=seed)
    sd["cost_regularization.prob.weight"] = sd["cost_regularization.prob.weight"] * gain
    net.load_state_dict(sd, strict=True)
    net.storage_dtype, net.num_depth = dtype, D
    net = net.cuda().eval()
    scene = synthetic.make_scene(1, V, H, W, seed=scene_seed)
    dev = {k: v.cuda() for k, v in scene.items()}
    taps = {}
    out = net(dev["imgs"], dev["K"], dev["R"], dev["t"], dev["depth_min"], dev["depth_max"], taps=taps)
    assert tuple(out["depth"].shape) == (1, H // 4, W // 4)
    planes = g["cost_planes"].tolist()
    bf = dtype == torch.bfloat16
    check_close(f"cfg1 cost volume {dtype}", taps["cost_volume"].float().permute(0, 4, 1, 2, 3)[:, :, planes].cpu(), t(g["cost_volume"]),
                rel_l2=2e-2 if bf else 3e-3)
    check_close(f"cfg1 logits {dtype}", taps["logits"].cpu(), t(g["logits"]), rel_l2=6e-2 if bf else 8e-3)
    ref = t(g["depth"])
    s = check_close(f"cfg1 depth {dtype}", out["depth"].cpu(), ref)
    if bf:   # the storage format's own cost is the yardstick (see tests/test_gpu_mvsnet.py)
        with torch.no_grad():
            emul = O.forward(scene["imgs"], scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"], sd,
                             num_depth=D, aggregation="softmin", store=dtype, store_feature_layers=True)["depth"]
        e = _rel_l1(emul, ref)
        print(f"[parity] cfg1 bf16: engine {s['rel_l1']:.3e} vs storage-emulated oracle {e:.3e}", flush=True)
        assert s["rel_l1"] <= 1.15 * e + 2e-5, (s, e)
    else:
        assert s["rel_l1"] <= 1e-3, s
    check_close(f"cfg1 confidence {dtype}", out["photometric_confidence"].cpu(), t(g["photometric_confidence"]), rel_l1=6e-2 if bf else 2e-2)


# ---------------------------------------------------------------------------------------------------------------------
# configuration (2): MVSNet (variance), 1 ref + 4 src, 512x640, D = 192 -- THE headline size, on windows against the oracle
# ---------------------------------------------------------------------------------------------------------------------
MVS_BF16_WINDOW_DEPTH = 1.25e-3
MVS_MARGIN = 30      # receptive radius of MVSNet's CostRegNet in the image plane (and along D: the depth axis is never cropped)


@pytest.mark.parametrize("rig", ["probe", "dtu"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_mvsnet_fullsize_matches_oracle_on_windows(gpu, dtype, rig):
    """5 x 512x640, D = 192 (h x w = 128 x 160, 3.9 M voxels): the size at which the LDS-staged warp kernel runs 32-plane chunks with
    four staged views, conv0 its 64-plane sweeps, the head its plane-ring sweep.  Both camera rigs of `synthetic.make_cameras`
    (the probe rig the bench runs and the DTU-like one: depth 425..905, tilted epipolar lines, 0.13-0.3 texels per plane).
    On two windows of reference pixels (image corner, interior), with the engine's own 16-bit feature maps as the oracle's input:
      * warp + variance: pointwise in the reference pixel -> the WHOLE window, every plane, to one rounding of the stored value;
      * logits (fp32 oracle U-Net on the engine's cost-volume window), depth and confidence: beyond the U-Net's +-30 pixel reach
        from the artificial window borders (sides that are the image border keep their zero padding and are compared to the edge)."""
    L, ops, synthetic = gpu
    from wild_deep_mvs_amd.models.MVSNet.model import MVSNet
    from oracle import mvsnet as O
    V, H, W, D = 5, 512, 640, 192
    net = MVSNet("variance")
    sd = synthetic.sharpened_state_dict("mvsnet", synthetic.template_of(net), seed=0)
    net.load_state_dict(sd, strict=True)
    net.storage_dtype, net.num_depth, net.graph_replay = dtype, D, False
    net = net.cuda().eval()
    scene = synthetic.make_scene(1, V, H, W, seed=2, rig=rig)
    dev = {k: v.cuda() for k, v in scene.items()}
    assert L.get_tuning("warp_tiled") == 1
    taps = {}
    with torch.no_grad():
        out = net(dev["imgs"], dev["K"], dev["R"], dev["t"], dev["depth_min"], dev["depth_max"], taps=taps)
        feats_cl = net.extract_features_cl([dev["imgs"][:, i] for i in range(V)])
    h, w = H // 4, W // 4
    assert tuple(out["depth"].shape) == (1, h, w) and torch.isfinite(out["depth"]).all()
    feats = [f.float().permute(0, 3, 1, 2).contiguous().cpu() for f in feats_cl]            # the engine's stored (16-bit) maps
    proj, dvals = O.mvsnet_cameras(scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"], D)
    dv = dvals[:, 0]
    cost = taps["cost_volume"].float().permute(0, 4, 1, 2, 3).cpu()                         # [1,32,D,h,w]
    logits = taps["logits"].cpu()                                                            # [1,D,h,w]
    ulp = 2 ** -8 if dtype == torch.bfloat16 else 2 ** -11
    win = 80
    bf = dtype == torch.bfloat16
    for (y0, x0) in ((0, 0), (24, 40), (h - win, w - win)):
        shift = torch.eye(4)
        shift[0, 2], shift[1, 2] = -float(x0), -float(y0)
        ref_proj = shift @ proj[:, 0]                                                        # a pinhole crop = a shifted principal point
        with torch.no_grad():
            warped = [O.homo_warping(feats[i], proj[:, i], ref_proj, dv, (win, win)) for i in range(1, V)]
            o_cost = O.variance_cost(feats[0][:, :, y0:y0 + win, x0:x0 + win].contiguous(), warped)
        e_cost = cost[:, :, :, y0:y0 + win, x0:x0 + win]
        s = check_close(f"cfg2 {rig} {dtype} cost volume window ({y0},{x0})", e_cost, o_cost, rel_l2=1.2 * ulp)
        assert s["max_abs"] <= 2 * ulp * s["ref_max"] + 1e-4, s
        with torch.no_grad():
            o_logits = O.cost_reg_net(e_cost.contiguous(), sd).squeeze(1)
            _, o_depth, o_conf = O.regress(o_logits, dv)
        ya = 0 if y0 == 0 else MVS_MARGIN
        xa = 0 if x0 == 0 else MVS_MARGIN
        yb = win if y0 + win == h else win - MVS_MARGIN
        xb = win if x0 + win == w else win - MVS_MARGIN
        sel = (slice(None), slice(None), slice(ya, yb), slice(xa, xb))
        check_close(f"cfg2 {rig} {dtype} logits window ({y0},{x0})", logits[:, :, y0:y0 + win, x0:x0 + win][sel], o_logits[sel],
                    rel_l2=2e-2 if bf else 3e-3)
        sel2 = sel[1:]
        e_depth = out["depth"].cpu()[:, y0:y0 + win, x0:x0 + win]
        sd_ = check_close(f"cfg2 {rig} {dtype} depth window ({y0},{x0})", e_depth[sel2], o_depth[sel2])
        # windows are 80 x 80 samples of the map: fp16 at the north-star bar; bf16 sits AT the bar on whole maps (asserted <= 1e-3 in
        # test_mvsnet_config2_end_to_end_depth_meets_1e3_in_fp16_and_bf16), single windows measured 4.4e-4 ... 1.05e-3 (round 4)
        assert sd_["rel_l1"] <= (MVS_BF16_WINDOW_DEPTH if bf else 1e-3), sd_
        e_conf = out["photometric_confidence"].cpu()[:, y0:y0 + win, x0:x0 + win]
        check_close(f"cfg2 {rig} {dtype} confidence window ({y0},{x0})", e_conf[sel2], o_conf[sel2], rel_l1=8e-2 if bf else 2e-2)


@pytest.mark.parametrize("rig", ["probe", "dtu"])
def test_mvsnet_config2_end_to_end_depth_meets_1e3_in_fp16_and_bf16(gpu, rig):
    """BASELINE.json's parity clause at BASELINE configuration 2 itself, whole maps, COMPOUNDED over every stage: 5-view 128 x 160 x 32
    feature maps (the resident inputs of bench.py's timed region), D = 192 -> depth, the engine in fp16 storage AND in bf16 storage
    (the format configuration 2 names) against the fp32 oracle's hot path on the same inputs: relative L1 <= 1e-3 for both.
    Measured in round 4 by bench.py only (fp16 8.7e-5, bf16 7.4e-4); a bf16 regression to 5e-3 would have passed the suite."""
    L, ops, synthetic = gpu
    from wild_deep_mvs_amd.models.MVSNet.model import MVSNet, build_proj_matrices
    from oracle import mvsnet as O
    V, H, W, D, C = 5, 512, 640, 192, 32
    h, w = H // 4, W // 4
    net = MVSNet("variance")
    sd = synthetic.sharpened_state_dict("mvsnet", synthetic.template_of(net), seed=0)
    net.load_state_dict(sd, strict=True)
    net.num_depth = D
    net = net.cuda().eval()
    cams = synthetic.make_cameras(1, V, H, W, rig=rig)
    Ks = cams["K"].clone()
    Ks[:, :, :2] /= 4
    proj = build_proj_matrices(Ks, cams["R"], cams["t"])
    steps = torch.arange(D, dtype=torch.float32).view(1, -1)
    dv = cams["depth_min"][:, :1] + (cams["depth_max"][:, :1] - cams["depth_min"][:, :1]) / (D - 1) * steps
    feats = synthetic.make_features(1, V, C, h, w, seed=5)
    with torch.no_grad():
        o_depth, _ = O.hot_path([feats[i] for i in range(V)], proj, dv.unsqueeze(1).expand(-1, V, -1), sd, streaming=True)
        rel = {}
        for dtype in (torch.float16, torch.bfloat16):
            net.storage_dtype = dtype
            fcl = [ops.to_channels_last(feats[i].cuda(), dtype) for i in range(V)]
            depth, _ = net.hot_path(fcl, proj.cuda(), dv.cuda().contiguous())
            assert torch.isfinite(depth).all()
            rel[dtype] = _rel_l1(depth.float().cpu(), o_depth)
    print(f"[parity] cfg2 {rig} end-to-end depth rel-L1 vs fp32 oracle: fp16 {rel[torch.float16]:.3e}, bf16 {rel[torch.bfloat16]:.3e}", flush=True)
    assert rel[torch.float16] <= 1e-3, rel
    assert rel[torch.bfloat16] <= 1e-3, rel


# ---------------------------------------------------------------------------------------------------------------------
# configurations (3) and (5): Vis-MVSNet at 5-view 512x640 [192,32,16] and 9-view 1152x1600 [256,32,16]
# ---------------------------------------------------------------------------------------------------------------------
VIS_CONFIGS = {
    3: dict(V=5, H=512, W=640, depth_nums=[192, 32, 16], interval_scales=[128 / 192, 1, 0.5]),
    5: dict(V=9, H=1152, W=1600, depth_nums=[256, 32, 16], interval_scales=[0.5, 1, 0.5]),
}
VIS_MARGIN = 16      # pair U-Net + head (8) then fuse U-Net + head (8) voxels of reach in the image plane


def _vis_net(synthetic, seed=0, dtype=torch.float16):
    from wild_deep_mvs_amd.models.VisMVSNet.frontend import Frontend
    net = Frontend()
    sd = synthetic.sharpened_state_dict("vis", synthetic.template_of(net), seed=seed)
    net.load_state_dict(sd, strict=True)
    net.storage_dtype = dtype
    return net.cuda().eval(), sd


def _nchw32(x, channels_last):
    x = x.float()
    return (x.permute(0, 3, 1, 2) if channels_last else x).contiguous().cpu()


# bf16 storage (the format BASELINE configuration 2 names) at the full sizes: the same windows, against the fp32 oracle.  Its
# 8-bit significand through ~20 stored tensors per stage does not meet the 1e-3 north-star bar on these nets (DESIGN.md section 5:
# no mixed scheme is possible without per-layer mixed-operand kernels); the bars below are 1.5 x the values measured in round 3
# and exist to catch regressions, fp16 keeps the north-star bar.  Measured (round 3): Vis configuration 3 stage depths 2.0e-3 / 4e-4 /
# 1.4e-4 (coarse -> fine), CVP configuration 4 levels 6e-4 ... 3e-5: the coarse stages carry the error, the FINAL maps are inside 1e-3
# (asserted below), like MVSNet's 7.4e-4 at configuration 2.
WINDOW_BARS = {torch.float16: dict(depth=1e-3, pair=2e-3, prob=3e-2, cvp_cost=3e-3, cvp_logits=2e-2, cvp_depth=1e-3),
               torch.bfloat16: dict(depth=6e-3, pair=1.2e-2, prob=2e-1, cvp_cost=2.5e-2, cvp_logits=1.5e-1, cvp_depth=6e-3)}


@pytest.mark.parametrize("cid,dtype", [(3, torch.float16), (5, torch.float16), (3, torch.bfloat16), (5, torch.bfloat16)])
def test_vis_fullsize_stages_match_oracle_on_windows(gpu, cid, dtype):
    """Every cascade stage of the full-size run against ``oracle.vismvsnet.single_stage`` on two windows of the stage's
    reference pixels (image corner and an interior window): fused depth, window probability and every pair depth."""
    L, ops, synthetic = gpu
    from oracle import vismvsnet as OV
    cfg = VIS_CONFIGS[cid]
    bars = WINDOW_BARS[dtype]
    net, sd = _vis_net(synthetic, dtype=dtype)
    scene = {k: v.cuda() for k, v in synthetic.make_scene(1, cfg["V"], cfg["H"], cfg["W"], seed=cid).items()}
    captured = []
    stages = (net.model.stage1, net.model.stage2, net.model.stage3)
    for st in stages:
        def wrapped(sample, depth_num, _orig=st.forward, **kw):
            out = _orig(sample, depth_num, **kw)
            captured.append((sample, depth_num, kw, out))
            return out
        st.forward = wrapped
    with torch.no_grad():
        out = net(scene["imgs"], scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"],
                  depth_nums=cfg["depth_nums"], interval_scales=cfg["interval_scales"])
    assert torch.isfinite(out["depth"]).all() and len(captured) == 3
    for k, ((ref_feat, ref_cam, srcs_feat, srcs_cam), depth_num, kw, (est, prob, pairs)) in enumerate(captured):
        cl = bool(stages[k]._channels_last_features)
        rf = _nchw32(ref_feat, cl)
        sf = [_nchw32(f, cl) for f in srcs_feat]
        n, c, h, w = rf.shape
        s_scale = kw["s_scale"]
        d_int = kw["depth_interval_override"].float().cpu()
        start = kw["depth_start_override"]
        start = ref_cam[:, 1:2, 3:4, 0:1].float().cpu() if start is None else start.float().cpu()
        rcam, scams = ref_cam.float().cpu(), [sc.float().cpu() for sc in srcs_cam]
        win = 64
        # window origins are multiples of 4: the stride-2 level of the U-Net must see the window in the phase it has in the image
        for (y0, x0) in ((0, 0), (4 * ((h - win) // 8), 4 * ((w - win) // 8) + (4 if w - win >= 8 else 0))):
            cam = rcam.clone()
            cam[:, 1, 0, 2] -= x0 * s_scale        # principal point of the cropped reference view (scaled by 1/s_scale inside)
            cam[:, 1, 1, 2] -= y0 * s_scale
            st_win = start if start.shape[-1] == 1 else start[:, :, y0:y0 + win, x0:x0 + win]
            with torch.no_grad():
                o_est, o_prob, o_pairs = OV.single_stage(rf[:, :, y0:y0 + win, x0:x0 + win].contiguous(), cam, sf, scams, sd,
                                                         f"model.stage{k + 1}", depth_num, st_win, d_int, s_scale)
            # compare away from the artificial window borders (sides on the image border keep their zero padding)
            ya, xa = (0 if y0 == 0 else VIS_MARGIN), (0 if x0 == 0 else VIS_MARGIN)
            yb, xb = win - VIS_MARGIN, win - VIS_MARGIN
            sel = (slice(None), slice(None), slice(ya, yb), slice(xa, xb))
            e_est = est.float().cpu()[:, :, y0:y0 + win, x0:x0 + win]
            s = check_close(f"cfg{cid} {dtype} stage{k + 1} depth window ({y0},{x0})", e_est[sel], o_est[sel])
            assert s["rel_l1"] <= bars["depth"], s
            if k == 2:          # the FINAL depth map meets the north-star bar in both storage formats at the full sizes
                assert s["rel_l1"] <= 1e-3, s
            check_close(f"cfg{cid} {dtype} stage{k + 1} prob window ({y0},{x0})", prob.float().cpu()[:, :, y0:y0 + win, x0:x0 + win][sel],
                        o_prob[sel], rel_l1=bars["prob"])
            m = VIS_MARGIN // 2
            ya, xa = (0 if y0 == 0 else m), (0 if x0 == 0 else m)
            selp = (slice(None), slice(None), slice(ya, win - m), slice(xa, win - m))
            for vi, ((ed, _), (o_ed, _)) in enumerate(zip(pairs, o_pairs)):
                sp = check_close(f"cfg{cid} {dtype} stage{k + 1} pair {vi} depth window ({y0},{x0})",
                                 ed.float().cpu()[:, :, y0:y0 + win, x0:x0 + win][selp], o_ed[selp])
                assert sp["rel_l1"] <= bars["pair"], sp


# ---------------------------------------------------------------------------------------------------------------------
# configuration (4): CVP-MVSNet, 5 views, 1024x1280, nscale = 5
# ---------------------------------------------------------------------------------------------------------------------
CVP_MARGIN = 20      # receptive radius of the CVP CostRegNet is 17 voxels


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_cvp_fullsize_levels_match_oracle_on_windows(gpu, dtype):
    """Every pyramid level of the full-size run (96 coarse planes, then 8 per-pixel hypotheses per level up to 1024x1280, where
    the 64-channel convs run on 10.5 M voxels): ``calDepthHypo`` on the whole map, warp + variance + U-Net + regression on windows."""
    L, ops, synthetic = gpu
    from wild_deep_mvs_amd.models.CVP_MVSNet.frontend import Frontend
    from wild_deep_mvs_amd.models.CVP_MVSNet.models.modules import conditionIntrinsics
    from oracle import cvpmvsnet as OC
    V, H, W, nscale = 5, 1024, 1280, 5
    net = Frontend()
    sd = synthetic.sharpened_state_dict("cvp", synthetic.template_of(net), seed=0)
    net.load_state_dict(sd, strict=True)
    net.storage_dtype = dtype
    bars = WINDOW_BARS[dtype]
    net = net.cuda().eval()
    scene = synthetic.make_scene(1, V, H, W, seed=4)
    scene["t"] = scene["t"] * 8
    dev = {k: v.cuda() for k, v in scene.items()}
    taps = {}
    with torch.no_grad():
        out = net(dev["imgs"], dev["K"], dev["R"], dev["t"], dev["depth_min"], dev["depth_max"], nscale=nscale, taps=taps)
        assert tuple(out["depth"].shape) == (1, H, W) and torch.isfinite(out["depth"]).all()
        # the engine's own pyramid (fp16, channels-last) as the oracle's input features
        pyr = net.model.featurePyramid.forward_engine(torch.cat([dev["imgs"][:, i] for i in range(V)], 0), nscale, dtype)
    feats = [[p[i:i + 1].float().permute(0, 3, 1, 2).contiguous().cpu() for p in pyr] for i in range(V)]     # [view][level] NCHW
    B = 1
    row = torch.tensor([0.0, 0.0, 0.0, 1.0])
    ref_ex = torch.cat((torch.cat((scene["R"][:, 0], scene["t"][:, 0]), 2), row.view(1, 1, 4).expand(B, 1, 4)), 1)
    src_ex = torch.cat((torch.cat((scene["R"][:, 1:], scene["t"][:, 1:]), 3), row.view(1, 1, 1, 4).expand(B, V - 1, 1, 4)), 2)
    shapes = [tuple(f.shape) for f in feats[0]]
    ref_in_ms = conditionIntrinsics(scene["K"][:, 0], scene["imgs"][:, 0].shape, shapes)               # [B,L,3,3]
    src_in_ms = [conditionIntrinsics(scene["K"][:, i], scene["imgs"][:, 0].shape, shapes) for i in range(1, V)]
    dmin, dmax = scene["depth_min"][:, 0], scene["depth_max"][:, 0]
    levels = [(nscale - 1, taps["coarse"])] + [(lv, lt) for lv, lt in zip(range(nscale - 2, -1, -1), taps["refine"])]
    depth_prev = None
    for level, lt in levels:
        hyp = lt["hypos"].float().cpu()
        h, w = shapes[level][2:]
        if hyp.dim() == 4:
            # per-pixel hypotheses of the engine (fp64 epipolar step + exact median on the GPU) against the oracle's calDepthHypo
            up = F.interpolate(depth_prev[None, :], size=None, scale_factor=2, mode="bicubic", align_corners=None).squeeze(0)
            o_hyp = OC.cal_depth_hypo(up, ref_in_ms[:, level], torch.stack([s[:, level] for s in src_in_ms], 1), ref_ex, src_ex, dmin, dmax)
            check_close(f"cfg4 level {level} hypotheses", hyp, o_hyp, max_abs=2e-5 * float(o_hyp.abs().max()))
        win = min(80, h)
        # window origins are multiples of 4 (stride-2 level of the U-Net: same phase as in the image)
        for (y0, x0) in ((0, 0), (4 * ((h - win) // 8), 4 * ((w - win) // 8))):
            rin = ref_in_ms[:, level].clone()
            rin[:, 0, 2] -= x0
            rin[:, 1, 2] -= y0
            hw_ = hyp if hyp.dim() == 2 else hyp[:, :, y0:y0 + win, x0:x0 + win].contiguous()
            ref_win = feats[0][level][:, :, y0:y0 + win, x0:x0 + win].contiguous()
            with torch.no_grad():
                warped = [OC.homo_warping(feats[i][level], rin, src_in_ms[i - 1][:, level], ref_ex, src_ex[:, i - 1], hw_, (win, win))
                          for i in range(1, V)]
                cost = OC.variance_cost(ref_win, warped)
                logits = OC.cost_reg_net(cost, sd)
                prob = torch.softmax(logits, 1)
                o_depth = torch.sum(prob * (hw_.view(B, -1, 1, 1) if hw_.dim() == 2 else hw_), 1)
            e_cost = lt["cost"].float().cpu()[:, :, y0:y0 + win, x0:x0 + win].permute(0, 4, 1, 2, 3)
            check_close(f"cfg4 {dtype} level {level} cost window ({y0},{x0})", e_cost, cost, rel_l2=bars["cvp_cost"])
            ya, xa = (0 if y0 == 0 else CVP_MARGIN), (0 if x0 == 0 else CVP_MARGIN)
            yb = win if y0 + win >= h else win - CVP_MARGIN
            xb = win if x0 + win >= w else win - CVP_MARGIN
            e_logits = lt["logits"].float().cpu()[:, :, y0:y0 + win, x0:x0 + win]
            check_close(f"cfg4 {dtype} level {level} logits window ({y0},{x0})", e_logits[:, :, ya:yb, xa:xb], logits[:, :, ya:yb, xa:xb], rel_l2=bars["cvp_logits"])
            e_depth = out["depth_est_list"][level].float().cpu()[:, y0:y0 + win, x0:x0 + win]
            s = check_close(f"cfg4 {dtype} level {level} depth window ({y0},{x0})", e_depth[:, ya:yb, xa:xb], o_depth[:, ya:yb, xa:xb])
            assert s["rel_l1"] <= bars["cvp_depth"], s
            if level == 0:      # the FINAL depth map meets the north-star bar in both storage formats
                assert s["rel_l1"] <= 1e-3, s
        depth_prev = out["depth_est_list"][level].float().cpu()


# ---------------------------------------------------------------------------------------------------------------------
# size-independent property at the full sizes: a sharded run returns the unsharded result
# (two ranks share cuda:0 and talk through gloo; the collectives, partial-sum kernels and LSE merges are the multi-GPU code)
# ---------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _vis_shard_worker(rank, world, port, cid, mode, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from wild_deep_mvs_amd import synthetic
        cfg = VIS_CONFIGS[cid]
        net, _ = _vis_net(synthetic)
        if world > 1:
            (net.set_depth_group if mode == "depth" else net.set_view_group)(dist.group.WORLD)
        scene = {k: v.cuda() for k, v in synthetic.make_scene(1, cfg["V"], cfg["H"], cfg["W"], seed=cid).items()}
        with torch.no_grad():
            out = net(scene["imgs"], scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"],
                      depth_nums=cfg["depth_nums"], interval_scales=cfg["interval_scales"])
        q.put((rank, world, out["depth"].float().cpu().numpy(), [d.float().cpu().numpy() for d in out["depth_est_list"]]))
    finally:
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("cid,mode", [(3, "depth"), (5, "view")])
@retry_infra
def test_vis_fullsize_shard_equals_unsharded(gpu, cid, mode):
    """BASELINE configuration (3) is the depth-plane shard, (5) the source-view shard: at the full size, two ranks return the
    depth map of the unsharded run (to the storage noise of the recomputed halo / the re-associated fused sum).  The view shard
    reduces 16-bit shares of the fused volume (round 3: reduce-scatter into slabs): the fused volume is a 16-bit sum of two 16-bit
    values instead of one rounded fp32 sum, which moves the stage depths by 2.6e-4 / 3.0e-4 at this size (the fp32 all-reduce of
    round 2: 1e-4); its bar is 5e-4, half the north-star's 1e-3."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    res = {}
    for world in (1, 2):
        port = _free_port()
        procs = [ctx.Process(target=_vis_shard_worker, args=(r, world, port, cid, mode, q)) for r in range(world)]
        for p in procs:
            p.start()
        got = [q.get(timeout=500) for _ in range(world)]
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0, f"a rank process exited with code {p.exitcode}"
        res[world] = sorted(got, key=lambda r: r[0])
    single = res[1][0]
    for rank, world, depth, ests in res[2]:
        rel = float(np.abs(depth - single[2]).mean() / np.abs(single[2]).mean())
        print(f"[parity] cfg{cid} {mode}-shard rank {rank}: depth rel-L1 vs unsharded {rel:.3e}", flush=True)
        bar = 5e-4 if mode == "view" else 3e-4
        assert rel <= bar, rel
        for a, b in zip(ests, single[3]):
            assert float(np.abs(a - b).mean() / np.abs(b).mean()) <= bar
    assert float(np.abs(res[2][0][2] - res[2][1][2]).max()) <= 1e-5, "ranks must agree"
