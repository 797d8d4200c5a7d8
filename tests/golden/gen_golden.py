#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/*.npz from the reference itself.

Runs ONLY in the build container (needs /root/reference, which never travels to
the GPU box).  The reference is imported as-is; its unused imports (cv2,
torchvision, h5py, matplotlib) are satisfied with empty stub modules and its
hard-coded ``.cuda()`` calls are neutralised by making ``Tensor.cuda`` /
``Module.cuda`` the identity in this process.

What is stored is data only: seeded synthetic inputs are NOT stored when they can
be re-created from ``wild_deep_mvs_amd.synthetic`` (the fixture records the
generator arguments instead); every stage boundary of the reference's hot path is
stored as fp32 arrays (tiny problem sizes; per-view warped volumes keep 3 planes).

Usage:  python tests/golden/gen_golden.py [--only mvsnet|mvsnet_s|mvsnet_dtu|mvsnet_train|mvsnet_s_train|vis|cvp|refframe|filter|photo|api]
"""
from __future__ import annotations

import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("PSCV_REFERENCE", "/root/reference")


def import_reference():
    sys.dont_write_bytecode = True
    for name in ["cv2", "torchvision", "torchvision.utils", "torchvision.transforms", "h5py", "matplotlib",
                 "matplotlib.pyplot"]:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
    tv = sys.modules["torchvision.transforms"]
    if not hasattr(tv, "ToPILImage"):
        tv.ToPILImage = lambda *a, **k: None
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    sys.path.insert(0, REF)


PLANES = [0, 5, 15]  # depth planes of the per-view warped volumes that are kept


def np32(x):
    return x.detach().to(torch.float32).cpu().numpy()


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


# --------------------------------------------------------------------------
def gen_mvsnet(aggregation: str, tag: str, *, H=64, W=96, V=3, D=16, seed=0, behind_view=-1, scene_seed=0, rig="probe"):
    sys.path.insert(0, REPO)
    from wild_deep_mvs_amd import synthetic
    from models.MVSNet.model import MVSNet  # reference
    from models.MVSNet.module import homo_warping, depth_regression  # reference

    torch.manual_seed(0)
    net = MVSNet(aggregation)
    net.num_depth = D
    sd = synthetic.sharpened_state_dict("mvsnet", synthetic.template_of(net), seed=seed)
    net.load_state_dict(sd, strict=True)
    net.eval()
    scene = synthetic.make_scene(1, V, H, W, seed=scene_seed, behind_view=behind_view, rig=rig)

    with torch.no_grad():
        out = net(scene["imgs"], scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"])
        # stage boundaries, re-run piecewise through the reference's own functions
        imgs = torch.unbind(scene["imgs"], 1)
        feats = net.extract_features(imgs)
        from utils.utils_3D import build_proj_matrices
        Ks = scene["K"].clone()
        Ks[:, :, :2] /= 4
        proj = build_proj_matrices(Ks, scene["R"], scene["t"])
        rng = torch.arange(D).view(1, 1, -1)
        step = (scene["depth_max"] - scene["depth_min"]) / (D - 1)
        depth_values = scene["depth_min"].unsqueeze(-1) + step.unsqueeze(-1) * rng
        dv = depth_values[:, 0]
        warped = [homo_warping(feats[i], proj[:, i], proj[:, 0], dv, feats[0].shape[-2:]) for i in range(1, V)]
        cost = net.build_cost_volume(feats[0], feats[1:], proj[:, 0], [proj[:, i] for i in range(1, V)], dv)
        reg = net.cost_regularization
        c0 = reg.conv0(cost)
        c1 = reg.conv1(c0)
        c2 = reg.conv2(c1)
        c3 = reg.conv3(c2)
        c4 = reg.conv4(c3)
        c5 = reg.conv5(c4)
        c6 = reg.conv6(c5)
        u7 = c4 + reg.conv7(c6)
        u9 = c2 + reg.conv9(u7)
        u11 = c0 + reg.conv11(u9)
        logits = reg.prob(u11)
        prob = torch.softmax(logits.squeeze(1), dim=1)
        # per-pixel depth planes through the reference warp (module.py:140-143)
        hh, ww = feats[0].shape[-2:]
        dpp = dv.view(1, D, 1, 1) * (1.0 + 0.05 * torch.rand(1, D, hh, ww, generator=torch.Generator().manual_seed(3)))
        warped_pp = homo_warping(feats[1], proj[:, 1], proj[:, 0], dpp, (hh, ww))

    print(f"[{tag}] max prob mean {prob.max(1)[0].mean():.3f}, logit std over D {logits.squeeze(1).std(1).mean():.3f}, "
          f"cost abs mean {cost.abs().mean():.4f}, depth range {out['depth'].min():.3f}..{out['depth'].max():.3f}")
    save(f"{tag}.npz",
         meta=np.array([H, W, V, D, seed, scene_seed, behind_view], dtype=np.int64), rig=np.array(rig),
         features=np.stack([np32(f) for f in feats]),
         proj=np32(proj), depth_values=np32(depth_values),
         warped_planes=np.array(PLANES, dtype=np.int64),
         warped=np.stack([np32(w_[:, :, PLANES]) for w_ in warped]).astype(np.float32),
         cost_volume=np32(cost),
         conv0=np32(c0), conv2=np32(c2), conv6=np32(c6), up7=np32(u7), up11=np32(u11),
         logits=np32(logits),
         depth=np32(out["depth"]), photometric_confidence=np32(out["photometric_confidence"]),
         depth_per_pixel=np32(dpp), warped_per_pixel=np32(warped_pp[:, :, PLANES]))


def gen_mvsnet_cfg1(tag: str = "mvsnet_s_cfg1", *, H=128, W=160, V=3, D=48, seed=1, scene_seed=4, prob_gain=4):
    """BASELINE.json configuration (1) at its real size: MVSNet-s (soft-min aggregation), 1 ref + 2 src views, 128x160 images,
    D = 48 -- the reference's own CPU-runnable case.  Stores the outputs and the logits whole (245 KB) and the cost volume on
    three planes only (it is 7.8 MB)."""
    sys.path.insert(0, REPO)
    from wild_deep_mvs_amd import synthetic
    from models.MVSNet.model import MVSNet  # reference
    torch.manual_seed(0)
    net = MVSNet("softmin")
    net.num_depth = D
    sd = synthetic.sharpened_state_dict("mvsnet", synthetic.template_of(net), seed=seed)
    sd["cost_regularization.prob.weight"] = sd["cost_regularization.prob.weight"] * prob_gain   # 48 planes: keep the softmax peaked
    net.load_state_dict(sd, strict=True)
    net.eval()
    scene = synthetic.make_scene(1, V, H, W, seed=scene_seed)
    with torch.no_grad():
        out = net(scene["imgs"], scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"])
        feats = net.extract_features(torch.unbind(scene["imgs"], 1))
        from utils.utils_3D import build_proj_matrices
        Ks = scene["K"].clone()
        Ks[:, :, :2] /= 4
        proj = build_proj_matrices(Ks, scene["R"], scene["t"])
        step = (scene["depth_max"] - scene["depth_min"]) / (D - 1)
        dv = (scene["depth_min"].unsqueeze(-1) + step.unsqueeze(-1) * torch.arange(D).view(1, 1, -1))[:, 0]
        cost = net.build_cost_volume(feats[0], feats[1:], proj[:, 0], [proj[:, i] for i in range(1, V)], dv)
        logits = net.cost_regularization(cost).squeeze(1)
        prob = torch.softmax(logits, dim=1)
    planes = [0, 17, 47]
    print(f"[{tag}] max prob mean {prob.max(1)[0].mean():.3f}, depth range {out['depth'].min():.3f}..{out['depth'].max():.3f}")
    save(f"{tag}.npz", meta=np.array([H, W, V, D, seed, scene_seed, prob_gain], dtype=np.int64), cost_planes=np.array(planes, dtype=np.int64),
         cost_volume=np32(cost[:, :, planes]), logits=np32(logits), depth=np32(out["depth"]),
         photometric_confidence=np32(out["photometric_confidence"]))


def gen_mvsnet_train(aggregation: str, tag: str, *, H=64, W=96, V=3, D=16, seed=0, scene_seed=0, B=2):
    """One training step of the reference in train() mode: forward with batch-statistics BatchNorm, the supervised L1 loss
    of models/trainer.py:163-167, backward.  Stores depth, loss, every parameter gradient's norm, full gradients of a
    subset, and the BatchNorm running statistics after the step."""
    sys.path.insert(0, REPO)
    from wild_deep_mvs_amd import synthetic
    from models.MVSNet.model import MVSNet  # reference

    torch.manual_seed(0)
    net = MVSNet(aggregation)
    net.num_depth = D
    sd = synthetic.train_state_dict("mvsnet", synthetic.template_of(net), seed=seed)
    net.load_state_dict(sd, strict=True)
    net.train()
    scene = synthetic.make_scene(B, V, H, W, seed=scene_seed)
    out = net(scene["imgs"], scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"])
    depth = out["depth"]
    gt, mask = synthetic.train_target(scene, depth.shape[1], depth.shape[2])
    loss = synthetic.supervised_loss(depth, gt, mask, scene["depth_min"], scene["depth_max"])
    loss.backward()
    grads = {k: p.grad for k, p in net.named_parameters()}
    norms = {k: float(g.norm()) for k, g in grads.items()}
    keep = [k for k in grads if k.startswith("cost_regularization.") and (".bn." in k or ".1." in k or "prob" in k or "conv0." in k
                                                                        or "conv1." in k or "conv6." in k or "conv11." in k)]
    keep += ["feature.conv0.conv.weight", "feature.feature.weight", "feature.feature.bias"] + (["temp"] if "temp" in grads else [])
    stats = {k: v for k, v in net.state_dict().items() if "running_" in k and k.startswith("cost_regularization.")}
    print(f"[{tag}] loss {float(loss):.5f}, depth range {depth.min():.3f}..{depth.max():.3f}, "
          f"|g conv0| {norms['cost_regularization.conv0.conv.weight']:.3e}, |g feature.conv0| {norms['feature.conv0.conv.weight']:.3e}")
    arrays = {"meta": np.array([H, W, V, D, seed, scene_seed, B], dtype=np.int64), "depth": np32(depth), "loss": np.float32(float(loss)),
              "norm_keys": np.array(list(norms.keys())), "norm_vals": np.array(list(norms.values()), dtype=np.float64)}
    for k in keep:
        arrays["grad:" + k] = np32(grads[k])
    for k, v in stats.items():
        arrays["stat:" + k] = np32(v)
    save(f"{tag}.npz", **arrays)


def gen_vis(tag, *, H=64, W=96, V=3, depth_nums=(16, 8, 4), interval_scales=(8.0, 4.0, 2.0), seed=0, scene_seed=0):
    sys.path.insert(0, REPO)
    from wild_deep_mvs_amd import synthetic
    from models.VisMVSNet.frontend import Frontend  # reference
    import models.VisMVSNet.model_cas as MC

    torch.manual_seed(0)
    net = Frontend()
    sd = synthetic.sharpened_state_dict("vis", synthetic.template_of(net), seed=seed)
    net.load_state_dict(sd, strict=True)
    net.eval()
    net.depth_nums = list(depth_nums)
    net.interval_scales = list(interval_scales)
    scene = synthetic.make_scene(1, V, H, W, seed=scene_seed)

    # capture stage-1 internals of the first source pair through the reference's own functions
    cap = {}
    orig_gc, orig_bcv = MC.groupwise_correlation, MC.SingleStage.build_cost_volume

    def gc(v1, v2, groups, dim):
        out = orig_gc(v1, v2, groups, dim)
        cap.setdefault("cost", []).append(out)
        return out

    def bcv(self, *a, **k):
        out = orig_bcv(self, *a, **k)
        cap.setdefault("warped", []).append(out)
        return out

    MC.groupwise_correlation, MC.SingleStage.build_cost_volume = gc, bcv
    hooks = []
    for si, st in enumerate([net.model.stage1, net.model.stage2, net.model.stage3]):
        hooks.append(st.reg.register_forward_hook(lambda m, i, o, si=si: cap.setdefault(f"interm{si}", []).append(o)))
        hooks.append(st.reg_fuse.register_forward_hook(lambda m, i, o, si=si: cap.setdefault(f"fuse_in{si}", []).append(i[0]) or cap.setdefault(f"score{si}", []).append(o)))
    try:
        with torch.no_grad():
            out = net(scene["imgs"], scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"],
                      depth_nums=list(depth_nums), interval_scales=list(interval_scales))
            feats = [net.model.feat_ext(im) for im in torch.unbind(scene["imgs"], 1)]
    finally:
        MC.groupwise_correlation, MC.SingleStage.build_cost_volume = orig_gc, orig_bcv
        for h in hooks:
            h.remove()
    n_src = V - 1
    pm = out["photometric_confidence"]
    print(f"[{tag}] prob maps mean {pm.mean(dim=(0, 2, 3)).tolist()}, depth range {out['depth'].min():.3f}..{out['depth'].max():.3f}, "
          f"uncert stage1 {out['depth_pair_list'][2][0][1][0].mean():.3f}")
    arrays = dict(
        meta=np.array([H, W, V, seed, scene_seed] + list(depth_nums), dtype=np.int64),
        interval_scales=np.array(interval_scales, dtype=np.float32),
        depth=np32(out["depth"]), photometric_confidence=np32(pm),
        depth_est_list=np.array([0]),   # placeholder so the key order is stable
    )
    for i, d in enumerate(out["depth_est_list"]):
        arrays[f"depth_est_{i}"] = np32(d)
    for si, pr in enumerate(out["depth_pair_list"]):            # finest first: stage 3, 2, 1
        for vi, (ed, unc) in enumerate(pr):
            arrays[f"pair_depth_s{3 - si}_v{vi}"] = np32(ed)
            arrays[f"pair_uncert_s{3 - si}_v{vi}"] = np32(unc[0])
    for k in range(3):                                            # per-scale features of every view
        arrays[f"feat_s{k + 1}"] = np.stack([np32(f[k]) for f in feats])
    # stage-1 internals, first source view (cap lists are in call order: stage1 views..., stage2 ..., stage3 ...)
    arrays["s1_warped_v0"] = np32(cap["warped"][0])
    arrays["s1_cost_v0"] = np32(cap["cost"][0])
    arrays["s1_interm_v0"] = np32(cap["interm0"][0])
    arrays["s1_fused"] = np32(cap["fuse_in0"][0])
    arrays["s1_score"] = np32(cap["score0"][0])
    arrays["s3_cost_v1"] = np32(cap["cost"][2 * n_src + 1])
    arrays["s3_fused"] = np32(cap["fuse_in2"][0])
    arrays["s3_score"] = np32(cap["score2"][0])
    save(f"{tag}.npz", **arrays)


def gen_vis_train(tag, *, H=64, W=96, V=3, depth_nums=(16, 8, 4), interval_scales=(8.0, 4.0, 2.0), seed=0, scene_seed=0, B=2):
    """One training step of the reference's Vis-MVSNet in train() mode (batch-statistics BatchNorm in the 2-D extractor, the
    pair / fuse U-Nets and the uncertainty net; stages detached from each other), the supervised loss of models/trainer.py
    (fused L1 per stage + the Bayesian pair loss with the predicted log-uncertainties), backward."""
    sys.path.insert(0, REPO)
    from wild_deep_mvs_amd import synthetic
    from models.VisMVSNet.frontend import Frontend  # reference
    import models.VisMVSNet.model_cas as MC

    # UncertNet.forward (model_cas.py:93-98) does `out += x` on the output of a non-inplace ReLU.  torch 1.4 (the reference's
    # pin) differentiates that ReLU from its INPUT, so the in-place add is harmless there; torch >= 1.7 differentiates it from
    # its OUTPUT and refuses ("modified by an inplace operation").  Same values, out of place, for this process only:
    def uncert_forward(self, x):
        out = self.conv2(self.conv1(x))
        out = out + x
        return [conv(out) for conv in self.head_convs]
    MC.UncertNet.forward = uncert_forward

    torch.manual_seed(0)
    net = Frontend()
    sd = synthetic.sharpened_state_dict("vis", synthetic.template_of(net), seed=seed)
    net.load_state_dict(sd, strict=True)
    net.train()
    net.depth_nums = list(depth_nums)
    net.interval_scales = list(interval_scales)
    scene = synthetic.make_scene(B, V, H, W, seed=scene_seed)
    out = net(scene["imgs"], scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"])
    gt, mask = synthetic.train_target(scene, H // 2, W // 2)
    loss = synthetic.vis_supervised_loss(out, gt, mask, scene["depth_min"], scene["depth_max"], V)
    loss.backward()
    grads = {k: p.grad for k, p in net.named_parameters() if p.grad is not None}
    norms = {k: float(g.norm()) for k, g in grads.items()}
    keep = [k for k in grads if ("stage1.reg." in k or "stage3.reg_fuse." in k or "reg_pair" in k or "uncert_net.head" in k)]
    keep += ["model.feat_ext.init_conv.0.weight", "model.feat_ext.final_conv_3.weight"]
    stats = {k: v for k, v in net.state_dict().items() if "running_" in k and ".stage" in k}
    print(f"[{tag}] loss {float(loss.detach()):.5f}, depth range {out['depth'].min():.3f}..{out['depth'].max():.3f}, params with grad "
          f"{len(grads)}/{len(list(net.parameters()))}, |g stage1 reg conv1| {norms['model.stage1.reg.unet.enc_blocks.reg14_0.0.conv1.weight']:.3e}")
    arrays = {"meta": np.array([H, W, V, seed, scene_seed, B], dtype=np.int64), "depth_nums": np.array(depth_nums, dtype=np.int64),
              "interval_scales": np.array(interval_scales, dtype=np.float64), "loss": np.float32(float(loss.detach())),
              "norm_keys": np.array(list(norms.keys())), "norm_vals": np.array(list(norms.values()), dtype=np.float64)}
    for i, d in enumerate(out["depth_est_list"]):
        arrays[f"depth_est_{i}"] = np32(d)
    for i, prs in enumerate(out["depth_pair_list"]):
        for j, (dp, (unc,)) in enumerate(prs):
            arrays[f"pair_{i}_{j}_depth"] = np32(dp)
            arrays[f"pair_{i}_{j}_uncert"] = np32(unc)
    for k in keep:
        arrays["grad:" + k] = np32(grads[k])
    for k, v in stats.items():
        arrays["stat:" + k] = np32(v)
    save(f"{tag}.npz", **arrays)


def gen_cvp(tag, *, H=32, W=48, V=3, nscale=2, seed=0, scene_seed=0, baseline_scale=8, head_mult=1):
    sys.path.insert(0, REPO)
    from wild_deep_mvs_amd import synthetic
    from models.CVP_MVSNet.frontend import Frontend  # reference
    import models.CVP_MVSNet.models.net as NET

    torch.manual_seed(0)
    net = Frontend()
    sd = synthetic.sharpened_state_dict("cvp", synthetic.template_of(net), seed=seed)
    # head_mult > 1 ("cvp_peaked"): the shared 1-channel head scaled up until the COARSE level's softmax over 96 planes is peaked too
    # (with the default gain its mean max-probability is 0.05: the coarse depth hardly depends on the volume; round-5 review)
    for k in list(sd.keys()):
        if k.endswith("prob0.weight"):
            sd[k] = sd[k] * head_mult
    net.load_state_dict(sd, strict=True)
    net.eval()
    net.model.nscale = nscale
    scene = synthetic.make_scene(1, V, H, W, seed=scene_seed)
    # wider baseline: calDepthHypo's interval (depth change per source pixel) stays a fraction of the depth range
    scene["t"] = scene["t"] * baseline_scale
    cap = {}
    orig_reg = net.model.cost_reg_refine.forward
    orig_hypo, orig_pc = NET.calDepthHypo, NET.proj_cost

    def reg(x):
        cap.setdefault("cost", []).append(x.clone())
        out = orig_reg(x)
        cap.setdefault("logits", []).append(out)
        return out

    def hypo(*a, **k):
        out = orig_hypo(*a, **k)
        cap.setdefault("hypos", []).append(out)
        return out

    net.model.cost_reg_refine.forward = reg
    NET.calDepthHypo = hypo
    try:
        with torch.no_grad():
            out = net(scene["imgs"], scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"], nscale=nscale)
            pyr = [net.model.featurePyramid(im, nscale) for im in torch.unbind(scene["imgs"], 1)]
    finally:
        net.model.cost_reg_refine.forward = orig_reg
        NET.calDepthHypo = orig_hypo
    p0 = torch.softmax(cap["logits"][0], 1)
    print(f"[{tag}] coarse max prob mean {p0.max(1)[0].mean():.3f}, refine max prob {torch.softmax(cap['logits'][-1], 1).max(1)[0].mean():.3f}, "
          f"depth range {out['depth'].min():.3f}..{out['depth'].max():.3f}, conf mean {out['photometric_confidence'].mean():.3f}")
    arrays = dict(meta=np.array([H, W, V, nscale, seed, scene_seed, baseline_scale], dtype=np.int64), head_mult=np.int64(head_mult),
                  coarse_max_prob_mean=np.float32(p0.max(1)[0].mean()),
                  depth=np32(out["depth"]), photometric_confidence=np32(out["photometric_confidence"]),
                  coarse_planes=np.array(PLANES96, dtype=np.int64),
                  coarse_cost=np32(cap["cost"][0][:, :, PLANES96]), coarse_logits=np32(cap["logits"][0]))
    for i, d in enumerate(out["depth_est_list"]):
        arrays[f"depth_est_{i}"] = np32(d)
    for lvl in range(nscale):
        arrays[f"pyr_l{lvl}"] = np.stack([np32(p[lvl]) for p in pyr])
    for i in range(1, len(cap["cost"])):
        arrays[f"refine{i}_hypos"] = np32(cap["hypos"][i - 1])
        arrays[f"refine{i}_cost"] = np32(cap["cost"][i])
        arrays[f"refine{i}_logits"] = np32(cap["logits"][i])
    save(f"{tag}.npz", **arrays)


PLANES96 = [0, 40, 95]


def gen_refframe(tag="refframe_tiny"):
    """The three reference models called with a LIST of per-view images (test-mode loaders, data/md_yao.py:126) and
    `reference_frame != 0`: depth maps only.  MVSNet ref 1 of 3 views, Vis-MVSNet ref 2 of 4, CVP-MVSNet ref 1 of 3."""
    sys.path.insert(0, REPO)
    from wild_deep_mvs_amd import synthetic
    from models.MVSNet.model import MVSNet
    from models.VisMVSNet.frontend import Frontend as VisFrontend
    from models.CVP_MVSNet.frontend import Frontend as CvpFrontend
    arrays = {}
    torch.manual_seed(0)
    with torch.no_grad():
        net = MVSNet("variance"); net.num_depth = 16
        net.load_state_dict(synthetic.sharpened_state_dict("mvsnet", synthetic.template_of(net), seed=0), strict=True); net.eval()
        sc = synthetic.make_scene(1, 3, 64, 96, seed=0)
        out = net([sc["imgs"][:, i] for i in range(3)], sc["K"], sc["R"], sc["t"], sc["depth_min"], sc["depth_max"], reference_frame=1)
        arrays["mvsnet_depth"], arrays["mvsnet_meta"] = np32(out["depth"]), np.array([64, 96, 3, 16, 0, 0, 1], dtype=np.int64)
        net = VisFrontend()
        net.load_state_dict(synthetic.sharpened_state_dict("vis", synthetic.template_of(net), seed=0), strict=True); net.eval()
        net.depth_nums, net.interval_scales = [16, 8, 4], [8.0, 4.0, 2.0]
        sc = synthetic.make_scene(1, 4, 64, 96, seed=6)
        out = net([sc["imgs"][:, i] for i in range(4)], sc["K"], sc["R"], sc["t"], sc["depth_min"], sc["depth_max"], reference_frame=2,
                  depth_nums=[16, 8, 4], interval_scales=[8.0, 4.0, 2.0])
        arrays["vis_depth"], arrays["vis_meta"] = np32(out["depth"]), np.array([64, 96, 4, 0, 6, 2], dtype=np.int64)
        net = CvpFrontend()
        net.load_state_dict(synthetic.sharpened_state_dict("cvp", synthetic.template_of(net), seed=0), strict=True); net.eval()
        net.model.nscale = 2
        sc = synthetic.make_scene(1, 3, 32, 48, seed=0)
        sc["t"] = sc["t"] * 8
        out = net([sc["imgs"][:, i] for i in range(3)], sc["K"], sc["R"], sc["t"], sc["depth_min"], sc["depth_max"], reference_frame=1, nscale=2)
        arrays["cvp_depth"], arrays["cvp_meta"] = np32(out["depth"]), np.array([32, 48, 3, 2, 0, 0, 8, 1], dtype=np.int64)
    print(f"[{tag}] depth ranges: mvsnet {arrays['mvsnet_depth'].min():.3f}..{arrays['mvsnet_depth'].max():.3f}, "
          f"vis {arrays['vis_depth'].min():.3f}..{arrays['vis_depth'].max():.3f}, cvp {arrays['cvp_depth'].min():.3f}..{arrays['cvp_depth'].max():.3f}")
    save(f"{tag}.npz", **arrays)


def gen_cvp_train(tag, *, H=64, W=96, V=3, nscale=2, seed=0, scene_seed=0, baseline_scale=8, B=2):
    """One training step of the reference's CVP-MVSNet in train() mode (48 coarse planes, fixed halving refinement
    intervals, batch-statistics BatchNorm, the regulariser called once per pyramid level), supervised loss over
    depth_est_list like models/trainer.py:118-167, backward."""
    sys.path.insert(0, REPO)
    from wild_deep_mvs_amd import synthetic
    from models.CVP_MVSNet.frontend import Frontend  # reference

    torch.manual_seed(0)
    net = Frontend()
    sd = synthetic.train_state_dict("cvp", synthetic.template_of(net), seed=seed)
    net.load_state_dict(sd, strict=True)
    net.train()
    net.model.nscale = nscale
    scene = synthetic.make_scene(B, V, H, W, seed=scene_seed)
    scene["t"] = scene["t"] * baseline_scale
    out = net(scene["imgs"], scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"], nscale=nscale)
    gt, mask = synthetic.train_target(scene, H, W)
    loss = synthetic.supervised_loss_list(out["depth_est_list"], gt, mask, scene["depth_min"], scene["depth_max"])
    loss.backward()
    grads = {k: p.grad for k, p in net.named_parameters()}
    norms = {k: float(g.norm()) for k, g in grads.items()}
    keep = [k for k in grads if "cost_reg_refine" in k and (".bn." in k or ".1." in k or "prob0" in k or "conv0." in k or "conv3." in k
                                                             or "conv5." in k or "conv6." in k)]
    keep += ["model.featurePyramid.conv0aa.0.weight", "model.featurePyramid.conv0bh.0.weight", "model.featurePyramid.conv0bh.0.bias"]
    stats = {k: v for k, v in net.state_dict().items() if "running_" in k}
    print(f"[{tag}] loss {float(loss):.5f}, depth range {out['depth'].min():.3f}..{out['depth'].max():.3f}, "
          f"|g conv0| {norms['model.cost_reg_refine.conv0.conv.weight']:.3e}, |g pyramid conv0aa| {norms['model.featurePyramid.conv0aa.0.weight']:.3e}")
    arrays = {"meta": np.array([H, W, V, nscale, seed, scene_seed, baseline_scale, B], dtype=np.int64), "loss": np.float32(float(loss)),
              "norm_keys": np.array(list(norms.keys())), "norm_vals": np.array(list(norms.values()), dtype=np.float64)}
    for i, d in enumerate(out["depth_est_list"]):
        arrays[f"depth_est_{i}"] = np32(d)
    for k in keep:
        arrays["grad:" + k] = np32(grads[k])
    for k, v in stats.items():
        arrays["stat:" + k] = np32(v)
    save(f"{tag}.npz", **arrays)


def gen_state_dict_keys():
    """Key order, names and shapes of the reference's state dicts (the checkpoint-compat contract)."""
    import json
    from models.MVSNet.model import MVSNet
    out = {}
    for tag, net in (("mvsnet", MVSNet("variance")), ("mvsnet_s", MVSNet("softmin"))):
        out[tag] = [[k, list(v.shape)] for k, v in net.state_dict().items()]
    try:
        from models.VisMVSNet.frontend import Frontend as Vis
        from models.CVP_MVSNet.frontend import Frontend as CVP
        out["vis"] = [[k, list(v.shape)] for k, v in Vis().state_dict().items()]
        out["cvp"] = [[k, list(v.shape)] for k, v in CVP().state_dict().items()]
    except Exception as e:  # pragma: no cover
        print("skipping vis/cvp key lists:", e)
    path = os.path.join(HERE, "state_dict_keys.json")
    json.dump(out, open(path, "w"), indent=0)
    print("wrote", path)


# --------------------------------------------------------------------------
def gen_filter(tag: str, *, V=5, H=48, W=64, seed=0, behind_view=-1, half_res_view=-1, near_view=-1, upsample=False,
               downscale=1, num_consistent=3):
    """Geometric-consistency filter: the reference's own ``evaluation.filtering.run`` on depth maps written to a
    scratch directory (its only interface is files + a dataloader batch).  Stores inputs and the three masks."""
    import tempfile
    from argparse import Namespace
    sys.path.insert(0, REPO)
    from wild_deep_mvs_amd import synthetic
    from evaluation import filtering as ref_filtering  # reference
    from evaluation.pipeline_utils import depth_folder_name  # reference

    sc = synthetic.make_filter_scene(V, H, W, seed=seed, behind_view=behind_view, half_res_view=half_res_view,
                                     near_view=near_view)
    depth, src = sc["depth"], sc["src_depth"]
    if upsample:   # the stored maps are at 1/downscale resolution and the filter upsamples them (filtering.py:54-58)
        depth = depth[::downscale, ::downscale].contiguous()
        src = [d[::downscale, ::downscale].contiguous() for d in src]
    with tempfile.TemporaryDirectory() as tmp:
        args = Namespace(model="m", nviews=V, data_path=tmp, scene="scene0", upsample=upsample, downscale=downscale,
                         max_reproj_error=1.0, depth_threshold=0.01, min_tri_angle=1.0, num_consistent=num_consistent,
                         debug=False)
        folder = os.path.join(tmp, "IntRes", "depthmaps", depth_folder_name(args), "scene0")
        os.makedirs(folder)
        np.savez(os.path.join(folder, "ref_out.npz"), depthmap=np32(depth))
        for i, d in enumerate(src):
            np.savez(os.path.join(folder, f"src{i}_out.npz"), depthmap=np32(d))
        batch = {"filename": ["ref"], "K": sc["K"].clone().unsqueeze(0), "R": sc["R"].clone().unsqueeze(0),
                 "t": sc["t"].clone().unsqueeze(0), "src_filenames": [[f"src{i}"] for i in range(V - 1)]}
        ref_filtering.tqdm = lambda it, **k: it
        ref_filtering.run([batch], args)
        out = np.load(os.path.join(tmp, "IntRes", "geometric_filtering", depth_folder_name(args), "scene0", "ref_out.npz"))
        masks = {k: out[k].copy() for k in ("mask_depth", "mask_disp", "geo_mask")}
    arrays = {"depth": np32(depth), "K": np32(sc["K"]), "R": np32(sc["R"]), "t": np32(sc["t"]),
              "meta": np.array([V, H, W, seed, behind_view, half_res_view, int(upsample), downscale, num_consistent, near_view]),
              "params": np.array([1.0, 0.01, 1.0], dtype=np.float32)}
    for i, d in enumerate(src):
        arrays[f"src_depth_{i}"] = np32(d)
    arrays.update(masks)
    save(f"{tag}.npz", **arrays)
    print({k: float(v.mean()) for k, v in masks.items()})


def gen_photo(tag: str, *, B=2, V=3, H=48, W=64, seed=0, behind_view=-1, masked=False, i_ref=0, geom_clamping=0.05):
    """Unsupervised photometric loss: the reference's own ``Trainer.photometricloss`` / ``masked_photometricloss``
    (models/trainer.py:221-278) on a ``Trainer`` built without its ``__init__`` (which only adds logging state and a
    ``.cuda()`` call); for the occlusion-masked variant ``dist.all_gather`` / ``dist.get_rank`` are replaced process-locally by
    stand-ins that hand over the other views' depth maps.  Stores inputs, the loss map, the mask, the scalar loss of
    trainer.py:169-174 and its gradient to the depth map."""
    from argparse import Namespace
    import torch.distributed as tdist
    sys.path.insert(0, REPO)
    from wild_deep_mvs_amd import synthetic
    from models.trainer import Trainer             # reference
    from utils.ssimLoss import SSIM                # reference
    from utils.utils_3D import build_proj_matrices  # reference

    sc = synthetic.make_photo_case(B, V, H, W, seed=seed, behind_view=behind_view)
    proj = build_proj_matrices(sc["K"].clone(), sc["R"], sc["t"])
    tr = Trainer.__new__(Trainer)
    tr.ims, tr.ssim, tr.args = {}, SSIM(), Namespace(geom_clamping=geom_clamping, occ_masking=masked)
    depth = sc["depths"][i_ref].clone().requires_grad_(True)
    if masked:
        saved = (tdist.all_gather, tdist.get_rank)
        def fake_all_gather(out_list, tensor, **k):
            for v in range(V):
                out_list[v] = sc["depths"][v].clone() if v != i_ref else tensor.detach().clone()
        tdist.all_gather, tdist.get_rank = fake_all_gather, (lambda *a, **k: i_ref)
        try:
            ssim, mask = tr.masked_photometricloss(sc["imgs"], depth, proj)
        finally:
            tdist.all_gather, tdist.get_rank = saved
    else:
        ssim, mask = tr.photometricloss(sc["imgs"], depth, proj)
    maskf = mask.float()
    loss = torch.sum(ssim * maskf) / torch.sum(maskf)
    loss.backward()
    save(f"{tag}.npz", meta=np.array([B, V, H, W, seed, behind_view, int(masked), i_ref]), geom_clamping=np.float32(geom_clamping),
         proj=np32(proj), ssim=np32(ssim), mask=np32(maskf), loss=np32(loss), grad_depth=np32(depth.grad))
    print(tag, "loss", float(loss), "mask mean", float(maskf.mean()), "grad abs mean", float(depth.grad.abs().mean()))


def gen_api_names():
    """Public names of the reference's ``models/utils.py`` and ``models/trainer.py`` (data only: a list of identifiers), so a
    CPU test can hold the drop-in to "every name the reference's callers can import exists" without the reference present."""
    import inspect
    import json
    import models.utils as RU          # reference
    import models.trainer as RT        # reference
    def public(mod):
        return sorted(n for n, v in vars(mod).items() if not n.startswith("_") and not inspect.ismodule(v))
    names = {"models.utils": public(RU),
             "models.trainer": public(RT),
             "models.trainer.Trainer": sorted(n for n in dir(RT.Trainer) if not n.startswith("__"))}
    path = os.path.join(HERE, "api_names.json")
    with open(path, "w") as f:
        json.dump(names, f, indent=1, sort_keys=True)
    print(f"wrote {path}: " + ", ".join(f"{k} {len(v)}" for k, v in names.items()))


def gen_homography(tag="homography_tiny"):
    """Function-level API of models/VisMVSNet/homography.py (rows A3 / A0 of the scope table): get_homographies with inv=True (planes
    uniform in inverse depth, :41-46) and with per-pixel depth; homography_warping with per-pixel matrices, its output and the
    gradient of sum(out * weight) with respect to `input` (grid_sample's autograd under the no_grad grid, :101-120)."""
    from models.VisMVSNet.homography import get_homographies, homography_warping
    sys.path.insert(0, REPO)
    from wild_deep_mvs_amd import synthetic
    n, V, h, w, hs, ws, c, d = 2, 2, 20, 28, 24, 36, 8, 6
    scene = synthetic.make_scene(n, V, h * 4, w * 4, seed=21)
    row = torch.tensor([0., 0., 0., 1.])

    def cam(v):
        ext = torch.cat((torch.cat((scene["R"][:, v], scene["t"][:, v]), 2), row.view(1, 1, 4).expand(n, 1, 4)), 1)
        intr = torch.zeros(n, 4, 4)
        intr[:, :3, :3] = scene["K"][:, v]
        intr[:, :2, :3] /= 4
        return torch.stack((ext, intr), 1)
    gen = torch.Generator().manual_seed(5)
    start = torch.full((n, 1, 1, 1), 2.0)
    interval = torch.full((n, 1, 1, 1), 0.6)
    H_inv = get_homographies(cam(0), cam(1), d, start, interval, inv=True)              # [n,d,1,1,3,3]
    H_lin = get_homographies(cam(0), cam(1), d, start, interval, inv=False)
    depth = 2.5 + 3.0 * torch.rand(n, 1, h, w, generator=gen)
    Hs = get_homographies(cam(0), cam(1), 1, depth, torch.zeros(n, 1, 1, 1))[:, 0].clone()   # [n,h,w,3,3]
    Hs[:, :4, :6] *= -1.0                                                                 # behind the source camera
    src = torch.randn(n, c, hs, ws, generator=gen).requires_grad_(True)
    wgt = torch.randn(n, c, h, w, generator=gen)
    out = homography_warping(src, Hs, (h, w))
    (out * wgt).sum().backward()
    save(tag + ".npz", left_cam=np32(cam(0)), right_cam=np32(cam(1)), depth_start=np32(start), depth_interval=np32(interval),
         depth_num=np.int64(d), H_inv=np32(H_inv), H_lin=np32(H_lin), pixel_depth=np32(depth), H_pixel=np32(Hs), src=np32(src),
         weight=np32(wgt), warped=np32(out), grad_src=np32(src.grad))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    import_reference()
    torch.set_num_threads(8)
    todo = {
        "mvsnet": lambda: gen_mvsnet("variance", "mvsnet_tiny"),
        "mvsnet_behind": lambda: gen_mvsnet("variance", "mvsnet_behind", V=4, behind_view=2, scene_seed=5),
        "mvsnet_s": lambda: gen_mvsnet("softmin", "mvsnet_s_tiny", seed=1),
        # the DTU-like rig (synthetic.make_cameras(rig="dtu"): depth 425..905, cameras on an arc, tilted epipolar lines), 5 views
        "mvsnet_dtu": lambda: gen_mvsnet("variance", "mvsnet_dtu_tiny", V=5, scene_seed=2, rig="dtu"),
        "mvsnet_cfg1": gen_mvsnet_cfg1,
        "mvsnet_train": lambda: gen_mvsnet_train("variance", "mvsnet_train"),
        "mvsnet_s_train": lambda: gen_mvsnet_train("softmin", "mvsnet_s_train", seed=1),
        "vis": lambda: gen_vis("vis_tiny"),
        "vis_train": lambda: gen_vis_train("vis_train"),
        "cvp": lambda: gen_cvp("cvp_tiny"),
        "cvp_peaked": lambda: gen_cvp("cvp_peaked", head_mult=int(os.environ.get("PSCV_CVP_HEAD_MULT", "4")), scene_seed=3),
        "cvp_train": lambda: gen_cvp_train("cvp_train"),
        "refframe": gen_refframe,
        "keys": gen_state_dict_keys,
        "filter": lambda: (gen_filter("filter_tiny", V=6, behind_view=4, half_res_view=3, near_view=2),
                           gen_filter("filter_upsample", V=4, seed=3, upsample=True, downscale=2, num_consistent=2)),
    }
    todo["api"] = gen_api_names
    todo["homography"] = gen_homography
    todo["photo"] = lambda: (gen_photo("photo_tiny"), gen_photo("photo_behind", V=4, behind_view=2, seed=3),
                             gen_photo("photo_masked", V=4, masked=True, i_ref=1, seed=5))
    for k, fn in todo.items():
        if args.only in (None, k):
            fn()


if __name__ == "__main__":
    main()
