"""HIP fused softmax / regression kernel vs the oracle."""
import numpy as np
import pytest
import torch

from _util import check_close, load_golden, t

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from wild_deep_mvs_amd import _lib as L, ops
    from oracle import mvsnet as O
    return L, ops, O


@pytest.mark.parametrize("fname", ["mvsnet_tiny.npz", "mvsnet_s_tiny.npz"])
def test_depth_and_confidence_from_reference_logits(env, fname):
    L, ops, O = env
    g = load_golden(fname)
    logits = t(g["logits"]).squeeze(1).contiguous()
    dv = t(g["depth_values"])[:, 0].contiguous()
    o = ops.softargmin(logits.cuda(), dv.cuda(), want_conf=True, want_prob=True, want_index=True)
    check_close("depth", o["depth"].cpu(), t(g["depth"]), max_abs=2e-5 * float(dv.max()))
    check_close("photometric confidence", o["conf"].cpu(), t(g["photometric_confidence"]), max_abs=2e-5)
    prob = torch.softmax(logits, 1)
    check_close("prob volume", o["prob"].cpu(), prob, max_abs=1e-6)


def test_per_pixel_planes_window_entropy_bf16(env):
    """Vis soft_argmin(window=2) / entropy semantics and per-pixel depth planes (CVP refine levels)."""
    L, ops, O = env
    gen = torch.Generator().manual_seed(0)
    B, D, h, w = 2, 16, 9, 13
    logits = torch.randn(B, D, h, w, generator=gen) * 3
    depth_pp = torch.rand(B, D, h, w, generator=gen) + torch.arange(D).view(1, D, 1, 1)
    o = ops.softargmin(logits.cuda(), depth_pp.cuda(), want_index=True, want_conf=True, conf_mode=1, window=2.0,
                       want_entropy=True)
    p = torch.softmax(logits, 1)
    idx = torch.arange(D, dtype=torch.float32).view(1, D, 1, 1)
    e_idx = (p * idx).sum(1)
    check_close("per-pixel depth", o["depth"].cpu(), (p * depth_pp).sum(1), max_abs=2e-5 * D)
    check_close("expected index", o["index"].cpu(), e_idx, max_abs=2e-5 * D)
    mask = ((idx - e_idx.unsqueeze(1)).abs() <= 2).float()
    check_close("window prob", o["conf"].cpu(), (p * mask).sum(1), max_abs=2e-5)
    check_close("entropy", o["entropy"].cpu(), (-p * p.clamp(1e-9, 1.0).log()).sum(1), max_abs=2e-5)
    # bf16 logits are read as bf16 and computed in fp32
    lb = logits.to(torch.bfloat16)
    ob = ops.softargmin(lb.cuda(), depth_pp.cuda())
    check_close("bf16 logits", ob["depth"].cpu(), (torch.softmax(lb.float(), 1) * depth_pp).sum(1), max_abs=2e-5 * D)


def test_shift_invariance_and_partials_at_full_size(env):
    """Properties at 192x128x160: softmax is invariant to a per-pixel logit shift, and the per-shard partials
    of two depth halves merge (log-sum-exp) into the full-range result -- the multi-GPU merge rule."""
    L, ops, O = env
    gen = torch.Generator().manual_seed(1)
    B, D, h, w = 1, 192, 128, 160
    logits = (torch.randn(B, D, h, w, generator=gen) * 4).cuda()
    dv = torch.linspace(2.0, 6.0, D).view(1, D).cuda()
    full = ops.softargmin(logits, dv, want_index=True)
    shifted = ops.softargmin(logits + 37.5, dv)
    assert float((full["depth"] - shifted["depth"]).abs().max()) <= 1e-4
    lo = ops.softargmin(logits[:, :96].contiguous(), dv[:, :96].contiguous(), want_partials=True)["partials"]
    hi = ops.softargmin(logits[:, 96:].contiguous(), dv[:, 96:].contiguous(), want_partials=True, index_offset=96)["partials"]
    m = torch.maximum(lo[:, 0], hi[:, 0])
    a, b = torch.exp(lo[:, 0] - m), torch.exp(hi[:, 0] - m)
    se = lo[:, 1] * a + hi[:, 1] * b
    depth = (lo[:, 2] * a + hi[:, 2] * b) / se
    index = (lo[:, 3] * a + hi[:, 3] * b) / se
    assert float((depth - full["depth"]).abs().max()) <= 1e-4
    assert float((index - full["index"]).abs().max()) <= 2e-3


@pytest.mark.parametrize("D", [1, 5, 8, 9, 16, 24, 32])
@pytest.mark.parametrize("per_pixel", [False, True])
def test_short_depth_axes_thread_per_pixel_kernel(env, D, per_pixel):
    """D <= 32 takes the one-thread-per-pixel kernel (CVP's 8 hypotheses, Vis stages 2-3): every output mode against the fp64
    formulas (model.py:207-215, nn_utils.py:453-470) and against the slice kernel (softargmin_small = 0); ragged pixel count."""
    L, ops, O = env
    gen = torch.Generator().manual_seed(D)
    B, h, w = 2, 19, 27
    logits = (torch.randn(B, D, h, w, generator=gen) * 4).cuda()
    depth = (torch.rand(B, D, h, w, generator=gen) + torch.arange(D).view(1, D, 1, 1)) if per_pixel else (torch.rand(B, D, generator=gen) * 3 + 1)
    depth = depth.cuda()
    p = torch.softmax(logits.double().cpu(), 1)
    idx = torch.arange(D, dtype=torch.float64).view(1, D, 1, 1)
    dd = depth.double().cpu() if per_pixel else depth.double().cpu().view(B, D, 1, 1)
    e_idx = (p * idx).sum(1)
    for mode in (0, 1):
        res = {}
        for small in (1, 0):
            L.set_tuning("softargmin_small", small)
            try:
                res[small] = ops.softargmin(logits, depth, want_index=True, want_conf=True, conf_mode=mode, window=2.0, want_entropy=True,
                                            index_offset=3)
            finally:
                L.set_tuning("softargmin_small", 1)
        o = res[1]
        check_close(f"depth D={D}", o["depth"].cpu(), (p * dd).sum(1).float(), max_abs=3e-5 * float(dd.max()))
        check_close(f"index D={D}", o["index"].cpu(), (e_idx + 3).float(), max_abs=2e-5 * max(D, 4))
        check_close(f"entropy D={D}", o["entropy"].cpu(), (-p * p.clamp(1e-9, 1.0).log()).sum(1).float(), max_abs=2e-5)
        for k in ("depth", "index", "entropy", "conf"):
            check_close(f"{k} small vs slice kernel D={D} mode {mode}", o[k].cpu(), res[0][k].cpu(), max_abs=3e-5 * max(1.0, float(res[0][k].abs().max())))
