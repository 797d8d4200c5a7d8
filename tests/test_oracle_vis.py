"""The Vis-MVSNet CPU oracle against outputs of the reference itself (tests/golden/vis_tiny.npz)."""
import numpy as np
import pytest
import torch

from _util import load_golden, t
from oracle import vismvsnet as OV
from wild_deep_mvs_amd import synthetic


def _template():
    import json, os
    from _util import GOLDEN
    keys = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))["vis"]
    from collections import OrderedDict
    return OrderedDict((k, tuple(s)) for k, s in keys)


def test_vis_forward_and_stage_boundaries():
    g = load_golden("vis_tiny.npz")
    H, W, V, seed, scene_seed = [int(x) for x in g["meta"][:5]]
    depth_nums = [int(x) for x in g["meta"][5:8]]
    scales = [float(x) for x in g["interval_scales"]]
    sd = synthetic.sharpened_state_dict("vis", _template(), seed=seed)
    scene = synthetic.make_scene(1, V, H, W, seed=scene_seed)
    taps = {}
    with torch.no_grad():
        out = OV.forward(scene["imgs"], scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"], sd,
                         depth_nums=depth_nums, interval_scales=scales, attr_interval_scales=scales, taps=taps)

    def close(name, got, ref, tol=2e-5):
        got = got.numpy() if isinstance(got, torch.Tensor) else got
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        err = np.abs(got - ref).max()
        assert err <= tol * max(1.0, np.abs(ref).max()), f"{name}: max err {err}"

    for k in range(3):
        feats = torch.stack([taps["features_ref"][k]] + [f[k] for f in taps["features_src"]])
        close(f"feat_s{k + 1}", feats, g[f"feat_s{k + 1}"], 5e-5)
    s1, s3 = taps["stages"][0], taps["stages"][2]
    close("s1_warped_v0", s1["warped0"], g["s1_warped_v0"], 5e-5)
    close("s1_cost_v0", s1["cost0"], g["s1_cost_v0"], 5e-5)
    close("s1_interm_v0", s1["interm0"], g["s1_interm_v0"], 5e-5)
    close("s1_fused", s1["fused"], g["s1_fused"], 5e-5)
    close("s1_score", s1["score"].unsqueeze(1), g["s1_score"], 5e-5)
    close("s3_fused", s3["fused"], g["s3_fused"], 1e-4)
    close("s3_score", s3["score"].unsqueeze(1), g["s3_score"], 1e-4)
    close("depth", out["depth"], g["depth"], 2e-5)
    close("photometric_confidence", out["photometric_confidence"], g["photometric_confidence"], 1e-4)
    for i in range(3):
        close(f"depth_est_{i}", out["depth_est_list"][i], g[f"depth_est_{i}"], 2e-5)
    for si, pr in enumerate(out["depth_pair_list"]):
        for vi, (ed, unc) in enumerate(pr):
            close(f"pair_depth_s{3 - si}_v{vi}", ed, g[f"pair_depth_s{3 - si}_v{vi}"], 5e-5)
            close(f"pair_uncert_s{3 - si}_v{vi}", unc[0], g[f"pair_uncert_s{3 - si}_v{vi}"], 1e-4)


def test_function_level_homographies_and_per_pixel_warp_match_the_reference():
    """tests/golden/homography_tiny.npz (written by the reference's models/VisMVSNet/homography.py): get_homographies with planes
    uniform in depth and in INVERSE depth (inv=True, :41-46), per-pixel matrices, homography_warping with per-pixel matrices and the
    gradient to its input -- the oracle, and the mirror's get_homographies (plain tensor math: runs on the CPU), against them."""
    g = load_golden("homography_tiny.npz")
    lc, rc = t(g["left_cam"]), t(g["right_cam"])
    d = int(g["depth_num"])
    start, interval = t(g["depth_start"]), t(g["depth_interval"])
    from wild_deep_mvs_amd.models.VisMVSNet.homography import get_homographies as mirror_h
    for inv, key in ((False, "H_lin"), (True, "H_inv")):
        want = t(g[key])
        for name, fn in (("oracle", OV.get_homographies), ("mirror", mirror_h)):
            got = fn(lc, rc, d, start, interval, inv=inv)
            assert tuple(got.shape) == tuple(want.shape)
            err = float((got - want).abs().max() / want.abs().max())
            print(f"[parity] {name} get_homographies(inv={inv}): max rel {err:.2e}")
            assert err <= 2e-6, (name, inv, err)
    # inverse-depth planes really differ from the linear ones in the interior and agree at both ends
    assert float((t(g["H_inv"])[:, 1:-1] - t(g["H_lin"])[:, 1:-1]).abs().max()) > 1e-3
    Hp = OV.get_homographies(lc, rc, 1, t(g["pixel_depth"]), torch.zeros(lc.shape[0], 1, 1, 1))[:, 0].clone()
    Hp[:, :4, :6] *= -1.0
    assert float((Hp - t(g["H_pixel"])).abs().max() / t(g["H_pixel"]).abs().max()) <= 2e-6
    src = t(g["src"]).clone().requires_grad_(True)
    out = OV.homography_warping(src, t(g["H_pixel"]), tuple(g["warped"].shape[2:]))
    (out * t(g["weight"])).sum().backward()
    assert float((out.detach() - t(g["warped"])).abs().max()) <= 1e-5
    assert float((src.grad - t(g["grad_src"])).abs().max()) <= 1e-5
