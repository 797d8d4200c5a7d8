"""The CPU oracle in train() mode against one training step of the reference itself (tests/golden/*_train.npz, made by
tests/golden/gen_golden.py): forward with batch-statistics BatchNorm, the supervised loss of models/trainer.py:163-167,
every gradient ATen autograd produces, and the running statistics after the step."""
import numpy as np
import pytest
import torch

from oracle import mvsnet as O
from wild_deep_mvs_amd import synthetic
from _util import check_close, load_golden, t


def oracle_train_step(agg, H, W, V, D, seed, scene_seed, B, store=None):
    from wild_deep_mvs_amd.models.MVSNet.model import MVSNet
    sd = synthetic.train_state_dict("mvsnet", synthetic.template_of(MVSNet(agg)), seed=seed)
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v) for k, v in sd.items()}
    scene = synthetic.make_scene(B, V, H, W, seed=scene_seed)
    stats = {}
    out = O.forward(scene["imgs"], scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"], sd, num_depth=D,
                    aggregation=agg, training=True, new_stats=stats, store=store)
    depth = out["depth"]
    gt, mask = synthetic.train_target(scene, depth.shape[1], depth.shape[2])
    loss = synthetic.supervised_loss(depth, gt, mask, scene["depth_min"], scene["depth_max"])
    loss.backward()
    grads = {k: v.grad for k, v in sd.items() if isinstance(v, torch.Tensor) and v.requires_grad}
    return depth.detach(), float(loss.detach()), grads, stats


@pytest.mark.parametrize("fname,agg", [("mvsnet_train.npz", "variance"), ("mvsnet_s_train.npz", "softmin")])
def test_oracle_train_step_matches_reference(fname, agg):
    g = load_golden(fname)
    H, W, V, D, seed, scene_seed, B = [int(x) for x in g["meta"]]
    depth, loss, grads, stats = oracle_train_step(agg, H, W, V, D, seed, scene_seed, B)
    check_close("depth", depth, t(g["depth"]), max_abs=2e-4)
    assert abs(loss - float(g["loss"])) <= 2e-4 * abs(float(g["loss"])), (loss, float(g["loss"]))
    for k, ref in zip(g["norm_keys"], g["norm_vals"]):
        got = float(grads[str(k)].norm())
        assert abs(got - ref) <= 2e-3 * ref + 1e-6, (k, got, ref)
    for k in g:
        if k.startswith("grad:"):
            check_close(k, grads[k[5:]], t(g[k]), rel_l2=2e-3)
        if k.startswith("stat:"):
            check_close(k, stats[k[5:]], t(g[k]), rel_l2=1e-4)


def cvp_oracle_train_step(H, W, V, nscale, seed, scene_seed, bscale, B):
    import json, os
    from collections import OrderedDict
    from _util import GOLDEN
    from oracle import cvpmvsnet as OC
    keys = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))["cvp"]
    sd = synthetic.train_state_dict("cvp", OrderedDict((k, tuple(s)) for k, s in keys), seed=seed)
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v) for k, v in sd.items()}
    scene = synthetic.make_scene(B, V, H, W, seed=scene_seed)
    scene["t"] = scene["t"] * bscale
    stats = {}
    out = OC.forward(scene["imgs"], scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"], sd, nscale=nscale,
                     training_hypos=True, training=True, new_stats=stats)
    gt, mask = synthetic.train_target(scene, H, W)
    loss = synthetic.supervised_loss_list(out["depth_est_list"], gt, mask, scene["depth_min"], scene["depth_max"])
    loss.backward()
    grads = {k: v.grad for k, v in sd.items() if isinstance(v, torch.Tensor) and v.requires_grad}
    return [d.detach() for d in out["depth_est_list"]], float(loss.detach()), grads, stats


def test_cvp_oracle_train_step_matches_reference():
    g = load_golden("cvp_train.npz")
    H, W, V, nscale, seed, scene_seed, bscale, B = [int(x) for x in g["meta"]]
    depths, loss, grads, stats = cvp_oracle_train_step(H, W, V, nscale, seed, scene_seed, bscale, B)
    for i, d in enumerate(depths):
        check_close(f"depth_est_{i}", d, t(g[f"depth_est_{i}"]), max_abs=3e-4)
    assert abs(loss - float(g["loss"])) <= 2e-4 * abs(float(g["loss"])), (loss, float(g["loss"]))
    for k, ref in zip(g["norm_keys"], g["norm_vals"]):
        got = float(grads[str(k)].norm())
        assert abs(got - ref) <= 3e-3 * ref + 1e-6, (k, got, ref)
    for k in g:
        if k.startswith("grad:"):
            check_close(k, grads[k[5:]], t(g[k]), rel_l2=3e-3)
        if k.startswith("stat:"):
            check_close(k, stats[k[5:]], t(g[k]), rel_l2=1e-4)


def vis_oracle_train_step(H, W, V, seed, scene_seed, B, depth_nums, interval_scales):
    import json, os
    from collections import OrderedDict
    from _util import GOLDEN
    from oracle import vismvsnet as OV
    keys = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))["vis"]
    sd = synthetic.sharpened_state_dict("vis", OrderedDict((k, tuple(s)) for k, s in keys), seed=seed)
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v) for k, v in sd.items()}
    scene = synthetic.make_scene(B, V, H, W, seed=scene_seed)
    stats = {}
    with OV.train_mode(stats):
        out = OV.forward(scene["imgs"], scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"], sd,
                         depth_nums=depth_nums, interval_scales=interval_scales, attr_interval_scales=interval_scales)
    gt, mask = synthetic.train_target(scene, H // 2, W // 2)
    loss = synthetic.vis_supervised_loss(out, gt, mask, scene["depth_min"], scene["depth_max"], V)
    loss.backward()
    grads = {k: v.grad for k, v in sd.items() if isinstance(v, torch.Tensor) and v.requires_grad and v.grad is not None}
    return out, float(loss.detach()), grads, stats


def test_vis_oracle_train_step_matches_reference():
    g = load_golden("vis_train.npz")
    H, W, V, seed, scene_seed, B = [int(x) for x in g["meta"]]
    out, loss, grads, stats = vis_oracle_train_step(H, W, V, seed, scene_seed, B, tuple(int(x) for x in g["depth_nums"]),
                                                    tuple(float(x) for x in g["interval_scales"]))
    for i, d in enumerate(out["depth_est_list"]):
        check_close(f"depth_est_{i}", d.detach(), t(g[f"depth_est_{i}"]), max_abs=3e-4)
    for i, prs in enumerate(out["depth_pair_list"]):
        for j, (dp, (unc,)) in enumerate(prs):
            check_close(f"pair_{i}_{j}_depth", dp.detach(), t(g[f"pair_{i}_{j}_depth"]), max_abs=3e-4)
            check_close(f"pair_{i}_{j}_uncert", unc.detach(), t(g[f"pair_{i}_{j}_uncert"]), max_abs=3e-4)
    assert abs(loss - float(g["loss"])) <= 2e-4 * abs(float(g["loss"])), (loss, float(g["loss"]))
    for k, ref in zip(g["norm_keys"], g["norm_vals"]):
        got = float(grads[str(k)].norm())
        assert abs(got - ref) <= 5e-3 * ref + 1e-5, (k, got, ref)
    for k in g:
        if k.startswith("grad:"):
            check_close(k, grads[k[5:]], t(g[k]), rel_l2=5e-3)
        if k.startswith("stat:"):
            check_close(k, stats[k[5:]], t(g[k]), rel_l2=1e-4)
