"""The CVP-MVSNet CPU oracle against outputs of the reference itself (tests/golden/cvp_tiny.npz), eval mode
(96 coarse planes, calDepthHypo refinement intervals)."""
import json
import os
from collections import OrderedDict

import numpy as np
import torch

from _util import GOLDEN, load_golden
from oracle import cvpmvsnet as OC
from wild_deep_mvs_amd import synthetic


def cvp_template():
    keys = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))["cvp"]
    return OrderedDict((k, tuple(s)) for k, s in keys)


def cvp_scene(g):
    H, W, V, nscale, seed, scene_seed, bscale = [int(x) for x in g["meta"]]
    scene = synthetic.make_scene(1, V, H, W, seed=scene_seed)
    scene["t"] = scene["t"] * bscale
    return scene, nscale, seed


def cvp_weights(g, template, seed):
    """The fixture's weights: ``sharpened_state_dict`` with the shared 1-channel head scaled by the fixture's ``head_mult``
    (cvp_peaked.npz: 4 -- the COARSE level's softmax over 96 planes is peaked there too, mean max-probability 0.29 against 0.05 in
    cvp_tiny.npz, so its depth check is not the near-vacuous case SURVEY section 8c warns about)."""
    sd = synthetic.sharpened_state_dict("cvp", template, seed=seed)
    mult = int(g["head_mult"]) if "head_mult" in g else 1
    if mult != 1:
        for k in list(sd.keys()):
            if k.endswith("prob0.weight"):
                sd[k] = sd[k] * mult
    return sd


import pytest


@pytest.mark.parametrize("fixture", ["cvp_tiny.npz", "cvp_peaked.npz"])
def test_cvp_forward_and_stage_boundaries(fixture):
    g = load_golden(fixture)
    scene, nscale, seed = cvp_scene(g)
    sd = cvp_weights(g, cvp_template(), seed)
    if fixture == "cvp_peaked.npz":
        assert float(g["coarse_max_prob_mean"]) >= 0.25, "the peaked fixture must have a peaked coarse softmax"
    taps = {}
    with torch.no_grad():
        out = OC.forward(scene["imgs"], scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"], sd,
                         nscale=nscale, taps=taps)

    def close(name, got, ref, tol=3e-5):
        got = got.numpy() if isinstance(got, torch.Tensor) else got
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        err = np.abs(got - ref).max()
        assert err <= tol * max(1.0, np.abs(ref).max()), f"{name}: max err {err}"

    for lvl in range(nscale):
        close(f"pyr_l{lvl}", torch.stack([taps["ref_pyr"][lvl]] + [p[lvl] for p in taps["src_pyrs"]]), g[f"pyr_l{lvl}"])
    planes = g["coarse_planes"].tolist()
    close("coarse_cost", taps["coarse_cost"][:, :, planes], g["coarse_cost"])
    close("coarse_logits", taps["coarse"]["logits"], g["coarse_logits"], 1e-4)
    for i, lt in enumerate(taps["refine"], start=1):
        close(f"refine{i}_hypos", lt["hypos"], g[f"refine{i}_hypos"], 1e-5)
        close(f"refine{i}_cost", lt["cost"], g[f"refine{i}_cost"])
        close(f"refine{i}_logits", lt["logits"], g[f"refine{i}_logits"], 1e-4)
    for i in range(nscale):
        close(f"depth_est_{i}", out["depth_est_list"][i], g[f"depth_est_{i}"], 2e-5)
    close("depth", out["depth"], g["depth"], 2e-5)
    close("photometric_confidence", out["photometric_confidence"], g["photometric_confidence"], 1e-4)
