"""The CPU restatement of the unsupervised photometric loss (oracle/photometric.py) against outputs and depth gradients of the
reference's own Trainer.photometricloss / masked_photometricloss (tests/golden/photo_*.npz)."""
import os

import numpy as np
import pytest
import torch

from oracle import photometric as P
from wild_deep_mvs_amd import synthetic

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load_case(tag):
    g = np.load(os.path.join(GOLD, f"{tag}.npz"))
    B, V, H, W, seed, behind, masked, i_ref = [int(v) for v in g["meta"]]
    sc = synthetic.make_photo_case(B, V, H, W, seed=seed, behind_view=behind)
    return g, sc, torch.from_numpy(g["proj"]), bool(masked), i_ref


@pytest.mark.parametrize("tag", ["photo_tiny", "photo_behind", "photo_masked"])
def test_oracle_photometric_loss_matches_reference(tag):
    g, sc, proj, masked, i_ref = load_case(tag)
    depth = sc["depths"][i_ref].clone().requires_grad_(True)
    if masked:
        ssim, mask = P.masked_photometricloss(sc["imgs"], depth, [sc["depths"][v] if v != i_ref else depth.detach()
                                                                   for v in range(sc["depths"].shape[0])], proj, i_ref,
                                              float(g["geom_clamping"]))
    else:
        ssim, mask, _ = P.photometricloss(sc["imgs"], depth, proj)
    loss = P.masked_mean_loss(ssim, mask)
    loss.backward()
    assert np.array_equal(mask.float().numpy(), g["mask"])
    assert np.abs(ssim.detach().numpy() - g["ssim"]).max() <= 1e-6
    assert abs(float(loss) - float(g["loss"])) <= 1e-6
    rel = np.abs(depth.grad.numpy() - g["grad_depth"]).sum() / np.abs(g["grad_depth"]).sum()
    print(f"[parity] {tag}: oracle vs reference grad_depth rel-L1 {rel:.2e}")
    assert rel <= 1e-5
    if tag == "photo_behind":
        assert (g["mask"] == 0).mean() > 0.3          # the behind-camera view is exercised
