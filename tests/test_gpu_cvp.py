"""CVP-MVSNet mirror on the HIP engine vs the reference golden (tests/golden/cvp_tiny.npz)."""
import numpy as np
import pytest
import torch

from _util import check_close, load_golden, t
from test_oracle_cvp import cvp_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from wild_deep_mvs_amd import _lib as L, ops, synthetic
    from wild_deep_mvs_amd.models.CVP_MVSNet.frontend import Frontend
    return L, ops, synthetic, Frontend


@pytest.mark.parametrize("feature_engine", ["pscv", "torch"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_cvp_forward_parity_with_reference(env, dtype, feature_engine):
    """Full forward vs the reference golden, with the 2-D pyramid tower on the HIP conv2d kernels (default) and on
    PyTorch-ROCm."""
    L, ops, synthetic, Frontend = env
    g = load_golden("cvp_tiny.npz")
    scene, nscale, seed = cvp_scene(g)
    net = Frontend()
    net.load_state_dict(synthetic.sharpened_state_dict("cvp", synthetic.template_of(net), seed=seed), strict=True)
    net.storage_dtype = dtype
    net.feature_engine = feature_engine
    net = net.cuda().eval()
    dev = {k: v.cuda() for k, v in scene.items()}
    taps = {}
    out = net(dev["imgs"], dev["K"], dev["R"], dev["t"], dev["depth_min"], dev["depth_max"], nscale=nscale, taps=taps)
    H, W = scene["imgs"].shape[-2:]
    assert tuple(out["depth"].shape) == (1, H, W) and tuple(out["photometric_confidence"].shape) == (1, 1, H, W)
    assert out["depth_pair_list"] == [] and len(out["depth_est_list"]) == nscale
    tol = 2e-3 if dtype == torch.float16 else 1.5e-2
    planes = g["coarse_planes"].tolist()
    check_close(f"coarse cost {dtype}", taps["coarse"]["cost"].float().permute(0, 4, 1, 2, 3)[:, :, planes].cpu(), t(g["coarse_cost"]), rel_l2=tol)
    check_close(f"coarse logits {dtype}", taps["coarse"]["logits"].cpu(), t(g["coarse_logits"]), rel_l2=3 * tol)
    for i, lt in enumerate(taps["refine"], start=1):
        check_close(f"refine{i} hypotheses {dtype}", lt["hypos"].cpu(), t(g[f"refine{i}_hypos"]), max_abs=0.02 if dtype == torch.float16 else 0.1)
        check_close(f"refine{i} logits {dtype}", lt["logits"].cpu(), t(g[f"refine{i}_logits"]), rel_l2=10 * tol)
    dtol = 1e-3 if dtype == torch.float16 else 5e-3
    for i in range(nscale):
        s = check_close(f"depth_est_list[{i}] {dtype}", out["depth_est_list"][i].cpu(), t(g[f"depth_est_{i}"]))
        assert s["rel_l1"] <= dtol, s
    check_close(f"confidence {dtype}", out["photometric_confidence"].cpu(), t(g["photometric_confidence"]), rel_l1=3e-2)


def test_cvp_function_level_warp_and_proj_cost(env):
    """models.CVP_MVSNet.models.modules.homo_warping / proj_cost drop-ins against the oracle (per-pixel hypotheses)."""
    L, ops, synthetic, Frontend = env
    from oracle import cvpmvsnet as OC
    from wild_deep_mvs_amd.models.CVP_MVSNet.models.modules import homo_warping, proj_cost
    g = load_golden("cvp_tiny.npz")
    scene, nscale, seed = cvp_scene(g)
    pyr = t(g["pyr_l0"])                                       # [V,1,16,H,W]
    B = 1
    row = torch.tensor([0., 0., 0., 1.])
    ref_ex = torch.cat((torch.cat((scene["R"][:, 0], scene["t"][:, 0]), 2), row.view(1, 1, 4).expand(B, 1, 4)), 1)
    src_ex = torch.cat((torch.cat((scene["R"][:, 1:], scene["t"][:, 1:]), 3), row.view(1, 1, 1, 4).expand(B, 2, 1, 4)), 2)
    hyp = t(g["refine1_hypos"])
    K = scene["K"]
    want = OC.homo_warping(pyr[1], K[:, 0], K[:, 1], ref_ex, src_ex[:, 0], hyp, pyr[0].shape[2:])
    got = homo_warping(pyr[1].cuda(), K[:, 0].cuda(), K[:, 1].cuda(), ref_ex.cuda(), src_ex[:, 0].cuda(), hyp.cuda(), pyr[0].shape[2:])
    check_close("cvp homo_warping per-pixel", got.cpu(), want, max_abs=3e-4)
    cost = proj_cost(2, pyr[0].cuda(), [[pyr[1].cuda()], [pyr[2].cuda()]], 0, K[:, 0].cuda(), K[:, 1:].cuda(), ref_ex.cuda(),
                     src_ex.cuda(), hyp.cuda(), storage_dtype=torch.float16)
    check_close("cvp proj_cost vs reference golden", cost.float().permute(0, 4, 1, 2, 3).cpu(), t(g["refine1_cost"]), rel_l2=2e-3)
