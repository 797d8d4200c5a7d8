"""CPU-side checks: the C-ABI library loads and exports every symbol of include/pscv.h, the conv3d weight
packer agrees with a numpy emulation of the kernel's contraction, host camera math, state-dict compatibility
with the reference.  No compute call into the GPU kernels here."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _util import GOLDEN, load_golden, t
from wild_deep_mvs_amd import _lib as L, ops, synthetic

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(REPO, "include", "pscv.h")).read()
    hdr_nc = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(pscv_[a-z0-9_]+)\s*\(", hdr_nc))
    assert declared == set(L.EXPORTS), (declared, L.EXPORTS)
    lib = C.CDLL(L.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"libpscv.so does not export {name}"
    assert L.lib().pscv_abi_version() == L.ABI_VERSION


def test_error_channel_and_argument_checks():
    lib = L.lib()
    rc = lib.pscv_set_tuning(b"no_such_knob", 1)
    assert rc != 0 and b"no_such_knob" in lib.pscv_last_error()
    # bad arguments are rejected before any launch
    rc = lib.pscv_softargmin(None, 0, None, 0, 0, None, None, None, None, None, None, 0, 2.0, 0, 1, 8, 4, 4, None)
    assert rc != 0 and b"logits" in lib.pscv_last_error()
    with pytest.raises(RuntimeError, match="no CPU"):
        ops.softargmin(torch.zeros(1, 8, 4, 4))


# ---- conv3d weight packing vs a numpy emulation of the kernel ------------------------------------
def _bf16(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(torch.bfloat16).to(torch.float32).numpy()


def _unpack(packed, nt, dtype=torch.bfloat16):
    if dtype == torch.bfloat16:
        w = (packed.astype(np.uint32) << 16).view(np.float32)
    else:
        w = packed.view(np.float16).astype(np.float32)
    return w.reshape(-1, nt, 64, 8)        # [step][tile][lane][j]


def _t2_class(pc):
    pd, ph, pw = (pc >> 2) & 1, (pc >> 1) & 1, pc & 1
    return pd, ph, pw, (1 + pd) * (1 + ph) * (1 + pw)


def _emulate(x, wk, cin, cout, kind):
    """x [D,H,W,cin] -> out [Do,Ho,Wo,cout]; follows conv3d.hip: for k-step s and lane group g the lane reads the
    8 channels (s*32+g*8)%cin.. of the voxel at tap (s*32+g*8)//cin; contraction over (g, j) against the packed A."""
    D, H, W, _ = x.shape
    nt = (cout + 15) // 16

    def X(d, h, w_, c0):
        if 0 <= d < D and 0 <= h < H and 0 <= w_ < W:
            return x[d, h, w_, c0:c0 + 8]
        return np.zeros(8, np.float32)

    if kind in (L.CONV_S1, L.CONV_S2):
        st = 1 if kind == L.CONV_S1 else 2
        Do, Ho, Wo = (D, H, W) if st == 1 else ((D + 1) // 2, (H + 1) // 2, (W + 1) // 2)
        out = np.zeros((Do, Ho, Wo, cout), np.float32)
        nsteps = (27 * cin + 31) // 32
        for od in range(Do):
            for oh in range(Ho):
                for ow in range(Wo):
                    for s in range(nsteps):
                        for g in range(4):
                            kk0 = s * 32 + g * 8
                            tap, c0 = min(kk0 // cin, 26), kk0 % cin
                            kd, kh, kw = tap // 9, (tap // 3) % 3, tap % 3
                            xv = X(od * st + kd - 1, oh * st + kh - 1, ow * st + kw - 1, c0)
                            for co in range(cout):
                                out[od, oh, ow, co] += wk[s, co // 16, (co % 16) + 16 * g] @ xv
        return out
    out = np.zeros((2 * D, 2 * H, 2 * W, cout), np.float32)
    sbase = 0
    for pc in range(8):
        pd, ph, pw, ntap = _t2_class(pc)
        nsteps = (ntap * cin + 31) // 32
        for id_ in range(D):
            for ih in range(H):
                for iw in range(W):
                    for s in range(nsteps):
                        for g in range(4):
                            kk0 = s * 32 + g * 8
                            tp, c0 = min(kk0 // cin, ntap - 1), kk0 % cin
                            tw, th, td = tp % (1 + pw), (tp // (1 + pw)) % (1 + ph), tp // ((1 + pw) * (1 + ph))
                            od = 1 if (pd and td == 0) else 0
                            oh = 1 if (ph and th == 0) else 0
                            ow = 1 if (pw and tw == 0) else 0
                            xv = X(id_ + od, ih + oh, iw + ow, c0)
                            for co in range(cout):
                                out[2 * id_ + pd, 2 * ih + ph, 2 * iw + pw, co] += wk[sbase + s, co // 16, (co % 16) + 16 * g] @ xv
        sbase += nsteps
    return out


@pytest.mark.parametrize("cin,cout,kind,transposed,dtype", [
    (8, 16, L.CONV_S1, False, torch.bfloat16), (32, 8, L.CONV_S1, False, torch.float16),
    (8, 1, L.CONV_S1, False, torch.bfloat16), (16, 32, L.CONV_S2, False, torch.bfloat16),
    (8, 16, L.CONV_S2, False, torch.float16), (16, 8, L.CONV_T2, True, torch.float16),
    (64, 32, L.CONV_T2, True, torch.bfloat16), (64, 32, L.CONV_S1, True, torch.bfloat16),
])
def test_packed_weights_contract_like_the_kernel(cin, cout, kind, transposed, dtype):
    rng = np.random.default_rng(cin * 100 + cout + kind)
    wshape = (cin, cout, 3, 3, 3) if transposed else (cout, cin, 3, 3, 3)
    w = _bf16(rng.standard_normal(wshape).astype(np.float32))     # bf16-representable values are exact in fp16 too
    D, H, W = (3, 2, 3) if kind != L.CONV_S2 else (3, 4, 5)
    x = _bf16(rng.standard_normal((D, H, W, cin)).astype(np.float32))
    packed = ops.pack_conv3d_weights(torch.from_numpy(w), kind, transposed, dtype)
    nt = (cout + 15) // 16
    got = _emulate(x, _unpack(packed, nt, dtype), cin, cout, kind)
    xt = torch.from_numpy(x).permute(3, 0, 1, 2).unsqueeze(0)
    wt = torch.from_numpy(w)
    if kind == L.CONV_T2:
        ref = F.conv_transpose3d(xt, wt, stride=2, padding=1, output_padding=1)
    elif transposed:
        ref = F.conv_transpose3d(xt, wt, stride=1, padding=1)
    else:
        ref = F.conv3d(xt, wt, stride=1 if kind == L.CONV_S1 else 2, padding=1)
    ref = ref[0].permute(1, 2, 3, 0).numpy()
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, atol=2e-4 * np.abs(ref).max(), rtol=0)


def test_sweep_pair_packing_contracts_like_the_kernel():
    """PSCV_CONV_S1P8 layout: rows 0-7 of the MFMA tile = output plane d, rows 8-15 = plane d+1; four input
    planes d-1..d+2 x nine (kh,kw) taps.  numpy emulation of conv3d_sweep.hip's contraction vs ATen."""
    rng = np.random.default_rng(5)
    cin, cout = 32, 8
    w = _bf16(rng.standard_normal((cout, cin, 3, 3, 3)).astype(np.float32))
    D, H, W = 5, 3, 4                              # odd D: the last pair's second plane is discarded
    x = _bf16(rng.standard_normal((D, H, W, cin)).astype(np.float32))
    packed = ops.pack_conv3d_weights(torch.from_numpy(w), L.CONV_S1P8, False, torch.float16)
    wk = packed.view(np.float16).astype(np.float32).reshape(4, 9, 64, 8)     # [p_rel][tap][lane][j]
    out = np.zeros((D + 1, H, W, cout), np.float32)
    for d in range(0, D, 2):
        for h in range(H):
            for x_ in range(W):
                for p in range(4):
                    for t_ in range(9):
                        kh, kw = t_ // 3, t_ % 3
                        pd, ph, pw = d - 1 + p, h + kh - 1, x_ + kw - 1
                        if not (0 <= pd < D and 0 <= ph < H and 0 <= pw < W):
                            continue
                        for g in range(4):
                            xv = x[pd, ph, pw, g * 8:g * 8 + 8]
                            for m in range(16):
                                out[d + (m >> 3), h, x_, m & 7] += wk[p, t_, m + 16 * g] @ xv
    ref = F.conv3d(torch.from_numpy(x).permute(3, 0, 1, 2).unsqueeze(0), torch.from_numpy(w), padding=1)[0]
    np.testing.assert_allclose(out[:D], ref.permute(1, 2, 3, 0).numpy(), atol=2e-4 * float(ref.abs().max()), rtol=0)


@pytest.mark.parametrize("cin,cout,transposed", [(8, 8, False), (16, 8, False), (16, 16, False), (8, 8, True), (16, 16, True)])
def test_narrow_sweep_packing_contracts_like_the_kernel(cin, cout, transposed):
    """PSCV_CONV_S1P8 layouts of the narrow depth-sweep kernel (conv3d_sweepc_kernel): the 32-deep MFMA reduction spans planes.
    c_in 8: lane group g = input plane d-1+g, one MFMA per tap; c_in 16 -> 8: two MFMA sets, plane 2 set + (g >> 1), channel half
    g & 1; rows 0-7 = output plane d, rows 8-15 = plane d+1.  16 -> 16: rows = the 16 channels of ONE output plane o, set 0 =
    planes (o-1, o), set 1 = (o, o+1) with zero weights on the repeated plane.  numpy emulation of the kernel's contraction
    against ATen, also for stride-1 ConvTranspose3d weights (flipped taps, swapped channel axes)."""
    rng = np.random.default_rng(cin + cout + transposed)
    wshape = (cin, cout, 3, 3, 3) if transposed else (cout, cin, 3, 3, 3)
    w = _bf16(rng.standard_normal(wshape).astype(np.float32))
    D, H, W = 5, 3, 4
    x = _bf16(rng.standard_normal((D, H, W, cin)).astype(np.float32))
    packed = ops.pack_conv3d_weights(torch.from_numpy(w), L.CONV_S1P8, transposed, torch.float16)
    nsets = 4 * cin // 32
    wk = packed.view(np.float16).astype(np.float32).reshape(nsets, 9, 64, 8)       # [set][tap][lane][j]
    out = np.zeros((D + 2, H, W, cout), np.float32)

    def plane_and_channels(set_, g, o):
        """input plane and channel slice lane group g of MFMA set `set_` reads while output plane(s) starting at o are computed"""
        if cout == 16:
            return o - 1 + set_ + (g >> 1), slice((g & 1) * 8, (g & 1) * 8 + 8)
        if cin == 8:
            return o - 1 + g, slice(0, 8)
        return o - 1 + 2 * set_ + (g >> 1), slice((g & 1) * 8, (g & 1) * 8 + 8)

    for o in range(0, D, 1 if cout == 16 else 2):
        for h in range(H):
            for x_ in range(W):
                for set_ in range(nsets):
                    for t_ in range(9):
                        kh, kw = t_ // 3, t_ % 3
                        ph, pw = h + kh - 1, x_ + kw - 1
                        if not (0 <= ph < H and 0 <= pw < W):
                            continue
                        for g in range(4):
                            pd, cs = plane_and_channels(set_, g, o)
                            if not 0 <= pd < D:
                                continue
                            xv = x[pd, ph, pw, cs]
                            for m in range(16):
                                if cout == 16:
                                    out[o, h, x_, m] += wk[set_, t_, m + 16 * g] @ xv
                                else:
                                    out[o + (m >> 3), h, x_, m & 7] += wk[set_, t_, m + 16 * g] @ xv
    xt = torch.from_numpy(x).permute(3, 0, 1, 2).unsqueeze(0)
    ref = (F.conv_transpose3d(xt, torch.from_numpy(w), padding=1) if transposed else F.conv3d(xt, torch.from_numpy(w), padding=1))[0]
    np.testing.assert_allclose(out[:D], ref.permute(1, 2, 3, 0).numpy(), atol=2e-4 * float(ref.abs().max()), rtol=0)


def test_parity_pair_deconv_packing_contracts_like_the_kernel():
    """PSCV_CONV_T2P8: numpy emulation of conv3d_t2p8.hip's contraction (9 k-steps, rows 0-7 / 8-15 = output x
    parity 0 / 1, K = two W taps x 16 channels) vs ATen's ConvTranspose3d(k3, s2, p1, op1)."""
    rng = np.random.default_rng(9)
    w = _bf16(rng.standard_normal((16, 8, 3, 3, 3)).astype(np.float32))
    D, H, W = 2, 3, 3
    x = _bf16(rng.standard_normal((D, H, W, 16)).astype(np.float32))
    wk = ops.pack_conv3d_weights(torch.from_numpy(w), L.CONV_T2P8, True, torch.float16).view(np.float16).astype(np.float32).reshape(9, 64, 8)
    out = np.zeros((2 * D, 2 * H, 2 * W, 8), np.float32)

    def X(d, h, w_, c0):
        return x[d, h, w_, c0:c0 + 8] if (d < D and h < H and w_ < W) else np.zeros(8, np.float32)

    for id_ in range(D):
        for ih in range(H):
            for iw in range(W):
                step = 0
                for pd in range(2):
                    for ph in range(2):
                        acc = np.zeros(16, np.float32)
                        for sd in range(pd + 1):
                            for sh in range(ph + 1):
                                od = 1 if (pd and sd == 0) else 0
                                oh = 1 if (ph and sh == 0) else 0
                                for g in range(4):
                                    xv = X(id_ + od, ih + oh, iw + (g >> 1), (g & 1) * 8)
                                    for m in range(16):
                                        acc[m] += wk[step, m + 16 * g] @ xv
                                step += 1
                        out[2 * id_ + pd, 2 * ih + ph, 2 * iw] = acc[:8]
                        out[2 * id_ + pd, 2 * ih + ph, 2 * iw + 1] = acc[8:]
    ref = F.conv_transpose3d(torch.from_numpy(x).permute(3, 0, 1, 2).unsqueeze(0), torch.from_numpy(w), stride=2, padding=1,
                             output_padding=1)[0].permute(1, 2, 3, 0).numpy()
    np.testing.assert_allclose(out, ref, atol=2e-4 * np.abs(ref).max(), rtol=0)


def test_one_channel_packing_layout():
    """PSCV_CONV_S1C1: depth-in-rows MFMA layout [step][lane][8] (conv3d_c1.hip).  Emulates the kernel's operand
    addressing in numpy -- A rows = 6 output planes, K = 8 input planes x 9 taps x c_in, step s = (q, tap), lane group
    g = lane >> 4 reads plane 4 (g >> 1) + 2 q + (g & 1) (c_in 8) or plane 4 (g >> 1) + q, channel half g & 1
    (c_in 16) -- and checks the result of D = A B against torch's conv3d."""
    import torch.nn.functional as F
    rng = np.random.default_rng(2)
    for cin in (8, 16):
        w = _bf16(rng.standard_normal((1, cin, 3, 3, 3)).astype(np.float32) / 8)
        packed = ops.pack_conv3d_weights(torch.from_numpy(w), L.CONV_S1C1, False, torch.float16)
        nsteps = 8 * 9 * cin // 32
        wk = packed.view(np.float16).astype(np.float32).reshape(nsteps, 64, 8)
        Dd, Hh, Ww = 6, 3, 16
        x = _bf16(rng.standard_normal((Dd, Hh, Ww, cin)).astype(np.float32))
        xp = np.zeros((Dd + 2, Hh + 2, Ww + 2, cin), np.float32)
        xp[1:-1, 1:-1, 1:-1] = x                                     # brick: plane p = d0 - 1 + p
        out = np.zeros((Dd, Hh, Ww), np.float32)
        for row in range(Hh):
            acc = np.zeros((16, 16), np.float32)                    # [m][n]
            for s in range(nsteps):
                A = np.zeros((16, 32), np.float32)
                Bm = np.zeros((32, 16), np.float32)
                for lane in range(64):
                    mn, g = lane & 15, lane >> 4
                    A[mn, g * 8:(g + 1) * 8] = wk[s, lane]
                    q, t = s // 9, s % 9
                    p = 4 * (g >> 1) + 2 * q + (g & 1) if cin == 8 else 4 * (g >> 1) + q
                    ch = slice(0, 8) if cin == 8 else slice(8 * (g & 1), 8 * (g & 1) + 8)
                    Bm[g * 8:(g + 1) * 8, mn] = xp[p, row + t // 3, mn + t % 3, ch]
                acc += A @ Bm
            out[:, row, :] = acc[:6]
            assert np.all(acc[6:] == 0)
        ref = F.conv3d(torch.from_numpy(x).permute(3, 0, 1, 2).unsqueeze(0), torch.from_numpy(w), padding=1)[0, 0].numpy()
        np.testing.assert_allclose(out, ref, atol=2e-4 * np.abs(ref).max(), rtol=0)


def test_fp16_packing_rounds_like_torch_and_saturates():
    """The host fp32->fp16 conversion used for the packed weights: round-to-nearest-even like torch's
    .to(float16), subnormals included, but SATURATING at +-65504 instead of overflowing to inf."""
    rng = np.random.default_rng(0)
    vals = np.concatenate([rng.standard_normal(4000) * 10.0 ** rng.integers(-9, 5, 4000),
                           [0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e6, -1e6, 6.1e-5, 6.0e-8, 2.9e-8, 3.1e-8]]).astype(np.float32)
    n = 27 * 8 * 8
    vals = np.resize(vals, ((vals.size + n - 1) // n) * n)
    got = []
    for blk in vals.reshape(-1, 8, 8, 3, 3, 3):                      # [co=8][ci=8][3][3][3]
        packed = ops.pack_conv3d_weights(torch.from_numpy(np.ascontiguousarray(blk)), L.CONV_S1, False, torch.float16)
        wk = packed.view(np.float16).reshape(-1, 1, 64, 8)            # [step][tile][lane][j]
        back = np.zeros_like(blk, dtype=np.float16)
        for co in range(8):
            for kk in range(27 * 8):
                s_, g, j = kk // 32, (kk % 32) // 8, kk % 8
                tap, ci = kk // 8, kk % 8
                back[co, ci, tap // 9, (tap // 3) % 3, tap % 3] = wk[s_, 0, co + 16 * g, j]
        got.append(back)
    got = np.stack(got).reshape(-1)
    want = torch.from_numpy(np.clip(vals, -65504.0, 65504.0)).to(torch.float16).numpy()
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))


def test_pack_rejects_bad_requests():
    w = torch.zeros(8, 8, 3, 3, 3)
    with pytest.raises(L.PscvError):
        ops.pack_conv3d_weights(w, L.CONV_T2, False)       # T2 needs a ConvTranspose3d weight
    with pytest.raises(ValueError):
        ops.pack_conv3d_weights(torch.zeros(8, 8, 1, 1, 1), L.CONV_S1, False)


# ---- host camera math ------------------------------------------------------------------------------
def test_proj_cams_matches_reference_matrix_product():
    g = load_golden("mvsnet_tiny.npz")
    proj = t(g["proj"])
    cams = ops.proj_cams([proj[:, 1], proj[:, 2]], proj[:, 0])
    assert tuple(cams.shape) == (2, 1, L.CAM_FLOATS)
    for i in (1, 2):
        P = (proj[:, i].double() @ torch.linalg.inv(proj[:, 0].double()))
        np.testing.assert_allclose(cams[i - 1, :, :9].reshape(-1, 3, 3).numpy(), P[:, :3, :3].numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(cams[i - 1, :, 9:12].numpy(), P[:, :3, 3].numpy(), rtol=1e-6, atol=1e-6)
        # and within fp32 LU noise of what the reference computes (module.py:128)
        P32 = proj[:, i] @ torch.inverse(proj[:, 0])
        np.testing.assert_allclose(cams[i - 1, :, :9].reshape(-1, 3, 3).numpy(), P32[:, :3, :3].numpy(), rtol=2e-5, atol=1e-6)


def test_bn_folding_equals_eval_batchnorm():
    torch.manual_seed(0)
    bn = torch.nn.BatchNorm3d(8).eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
    x = torch.randn(1, 8, 2, 3, 4)
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    bias = bn.bias - bn.running_mean * scale
    np.testing.assert_allclose((x * scale.view(1, -1, 1, 1, 1) + bias.view(1, -1, 1, 1, 1)).detach().numpy(),
                               bn(x).detach().numpy(), atol=1e-5)


# ---- drop-in contract ------------------------------------------------------------------------------
@pytest.mark.parametrize("arch,ctor", [("mvsnet", "variance"), ("mvsnet_s", "softmin")])
def test_state_dict_keys_and_shapes_match_reference(arch, ctor):
    ref = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))[arch]
    from wild_deep_mvs_amd.models.MVSNet.model import MVSNet
    mine = {k: list(v.shape) for k, v in MVSNet(ctor).state_dict().items()}
    assert list(mine.keys()) == [k for k, _ in ref], "state-dict key order/names differ from the reference"
    for k, shape in ref:
        assert mine[k] == shape, (k, mine[k], shape)


def test_vis_state_dict_keys_and_shapes_match_reference():
    ref = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))["vis"]
    from wild_deep_mvs_amd.models.VisMVSNet.frontend import Frontend
    net = Frontend()
    mine = [[k, list(v.shape)] for k, v in net.state_dict().items()]
    assert mine == ref
    assert net.depth_nums == [32, 16, 8] and net.interval_scales == [4, 2, 1]      # frontend.py:10-11


def test_cvp_state_dict_keys_and_shapes_match_reference():
    ref = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))["cvp"]
    from wild_deep_mvs_amd.models.CVP_MVSNet.frontend import Frontend
    net = Frontend()
    assert [[k, list(v.shape)] for k, v in net.state_dict().items()] == ref
    assert net.model.nscale == 2                                                   # net.py:92-93


def test_reference_import_paths_resolve():
    import wild_deep_mvs_amd
    wild_deep_mvs_amd.install_as_models()
    from models.MVSNet.model import MVSNet            # reference train.py:33
    from models.MVSNet.module import homo_warping, depth_regression   # noqa: F401
    from models.VisMVSNet.frontend import Frontend as Vis             # reference train.py:34
    from models.CVP_MVSNet.frontend import Frontend as CVP            # reference train.py:35
    from models.VisMVSNet.homography import homography_warping, get_homographies   # noqa: F401
    from models.VisMVSNet.nn_utils import soft_argmin, entropy, groupwise_correlation   # noqa: F401
    from models.CVP_MVSNet.models.modules import proj_cost            # noqa: F401
    from models.utils import homo_warp, rec_upsample, bayesian_version_loss   # noqa: F401  (BASELINE.json names models/utils.py homo_warp)
    from models.trainer import Trainer                                # noqa: F401  (loss half of the reference's trainer)
    assert homo_warp is homo_warping
    l, u, m = torch.rand(2, 1, 4, 5), torch.randn(2, 1, 4, 5), (torch.rand(2, 1, 4, 5) > 0.3).float()
    want = ((l * torch.exp(-u) + u) * m).sum() / m.sum() + (l * m).sum() / m.sum()          # models/utils.py:110-115
    assert torch.allclose(bayesian_version_loss(l, u, m), want)
    up = rec_upsample([torch.rand(2, 4, 5), (torch.rand(2, 1, 4, 5), None)], (8, 10))
    assert tuple(up[0].shape) == (2, 8, 10) and tuple(up[1][0].shape) == (2, 1, 8, 10) and up[1][1] is None
    assert Vis().depth_nums == [32, 16, 8] and CVP().model.nscale == 2
    net = MVSNet("variance")
    assert net.num_depth == 192
    with pytest.raises(RuntimeError, match="no CPU"):
        net.eval()(torch.rand(1, 3, 3, 32, 32), torch.eye(3).expand(1, 3, 3, 3).clone(), torch.eye(3).expand(1, 3, 3, 3).clone(),
                   torch.zeros(1, 3, 3, 1), torch.full((1, 3), 2.0), torch.full((1, 3), 6.0))


def test_install_as_models_exports_every_public_name_of_the_reference():
    """``install_as_models()`` replaces the whole ``models`` package, so ``models.utils`` and ``models.trainer`` must be
    SUPERSETS of the reference's modules: train.py calls ``trainer.step`` / ``trainer.log_iter``, depthmap_eval.py and
    evaluation/run_depthmaps.py do ``from models.utils import *`` and use ``tocuda`` / ``Thres_metrics``.  The name lists were
    dumped from the reference itself (tests/golden/gen_golden.py --only api -> api_names.json: identifiers only)."""
    import json
    import wild_deep_mvs_amd
    wild_deep_mvs_amd.install_as_models()
    import models.trainer as MT
    import models.utils as MU
    names = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "api_names.json")))
    assert [n for n in names["models.utils"] if not hasattr(MU, n)] == []
    assert [n for n in names["models.trainer"] if not hasattr(MT, n)] == []
    assert [n for n in names["models.trainer.Trainer"] if not callable(getattr(MT.Trainer, n, None))] == []
    # behaviour of the host-side helpers on nested containers / masks
    x = {"a": [torch.tensor(1.5), (torch.tensor(2.0),)], "b": 3.0}
    assert MU.tensor2float(x) == {"a": [1.5, (2.0,)], "b": 3.0}
    assert MU.tensor2numpy([torch.arange(3)])[0].tolist() == [0, 1, 2]
    assert MU.add_batch({"k": torch.zeros(2, 3), "name": "scan1"})["k"].shape == (1, 2, 3)
    est = torch.tensor([[[1.0, 2.0], [3.0, 8.0]], [[1.0, 1.0], [1.0, 1.0]]])
    gt = torch.tensor([[[1.0, 2.5], [3.0, 4.0]], [[2.0, 1.0], [1.0, 1.0]]])
    mask = torch.tensor([[[True, True], [True, True]], [[True, True], [False, False]]])
    assert abs(float(MU.AbsDepthError_metrics(est, gt, mask)) - ((0 + 0.5 + 0 + 4) / 4 + (1 + 0) / 2) / 2) < 1e-6
    assert abs(float(MU.Thres_metrics(est, gt, mask, 1)) - (1 / 4 + 0 / 2) / 2) < 1e-6
    assert abs(float(MU.Rel_Thres_metrics(est, gt, mask, 1.5)) - ((1 - 1 / 4) + (1 - 1 / 2)) / 2) < 1e-6
    assert abs(float(MU.RelDepthError_metrics(est, gt, mask)) - ((0.5 / 2.5 + 4 / 4) / 4 + (1 / 2) / 2) / 2) < 1e-6
    assert abs(float(MU.SquareRelDepthError_metrics(est, gt, mask)) - ((0.25 / 2.5 + 16 / 4) / 4 + (1 / 2) / 2) / 2) < 1e-6
    # the harness bookkeeping
    tr = MT.Trainer(model=None, args=type("A", (), {"print_every": 2, "architecture": "mvsnet", "upsample_training": False})())
    assert (tr.input_down, tr.output_down) == (1, 4)
    tr.keep_losses({"train_loss": torch.tensor(1.0)}); tr.keep_losses({"train_loss": torch.tensor(3.0), "val_loss": torch.tensor(5.0)})
    assert float(tr.log_iter()["train_loss"]) == 2.0 and tr.log_iter() == {}
    up = MT.Trainer(args=type("A", (), {"architecture": "cvp_mvsnet", "upsample_training": True})())
    assert (up.input_down, up.output_down) == (4, 1)
    # function-level geometry helpers pulled into models.trainer's namespace
    g = MT.build_grid(2, 3, torch.device("cpu"), normed=False)
    assert g.shape == (1, 2, 3, 2) and g[0, 1, 2].tolist() == [2, 1]
    n = MT.normalize(torch.tensor([[[0.0, 0.0], [4.0, 2.0]]]), 3, 5)
    assert torch.allclose(n, torch.tensor([[[-1.0, -1.0], [1.0, 1.0]]]))
    P = torch.eye(4).repeat(1, 2, 1, 1)
    P[0, 1, 0, 3] = 2.0        # second view: x shifted by 2 / depth
    flow, z = MT.flows_from_single_depthmap(torch.full((1, 2, 2), 2.0), P, 0)
    assert flow.shape == (1, 1, 2, 2, 2) and torch.allclose(flow[0, 0, 1, 1], torch.tensor([2.0, 1.0])) and torch.allclose(z, torch.full((1, 1, 2, 2), 2.0))


def test_train_layer_cache_keeps_one_entry_per_tag_across_optimiser_steps():
    """training._cached_layer: an optimiser step bumps ``weight._version``; the packed layers of the previous step must be
    evicted (round-2 advisor finding: the purge compared the epoch with the tag and never evicted -> unbounded growth)."""
    import torch.nn as nn
    from wild_deep_mvs_amd import ops, training as T
    net = nn.Module()
    ws = [nn.Parameter(torch.zeros(4, 4, 3, 3)) for _ in range(3)]
    made = []

    def use_all():
        for i, wt in enumerate(ws):
            T._cached_layer(net, f"f{i}", wt, torch.float16, lambda: made.append(1) or object())
            T._cached_layer(net, f"d{i}", wt, torch.float16, lambda: made.append(1) or object())
    use_all()
    assert len(net._pscv_train_layers) == 6 and len(made) == 6
    use_all()                                            # same versions: cache hits
    assert len(made) == 6
    for step in range(4):
        with torch.no_grad():
            for wt in ws:
                wt.add_(1.0)                             # what optimizer.step() does: in-place update -> _version + 1
        use_all()
        assert len(net._pscv_train_layers) == 6
    assert len(made) == 6 * 5
    ops.invalidate_weight_caches()                       # epoch bump: rebuilt once, still one entry per tag
    use_all()
    assert len(net._pscv_train_layers) == 6 and len(made) == 6 * 6


def test_tuning_knobs_are_process_wide_with_a_per_thread_override():
    """pscv_set_tuning reaches every host thread (autograd's backward thread, DataParallel replicas: round-2 advisor finding on
    the thread-local knobs); pscv_set_tuning_thread overrides for the calling thread only."""
    import threading
    seen = {}

    def other(tag):
        seen[tag] = L.get_tuning("c1_sweep")
    assert L.get_tuning("c1_sweep") == 1 and L.get_tuning("warp_tiled") == 1 and L.get_tuning("sweep_dc") == 0
    try:
        L.set_tuning("c1_sweep", 0)
        th = threading.Thread(target=other, args=("after_process_set",)); th.start(); th.join()
        assert seen["after_process_set"] == 0 and L.get_tuning("c1_sweep") == 0
        L.set_tuning_thread("c1_sweep", 2)
        th = threading.Thread(target=other, args=("after_thread_override",)); th.start(); th.join()
        assert L.get_tuning("c1_sweep") == 2 and seen["after_thread_override"] == 0
        L.set_tuning_thread("c1_sweep", 0, enable=False)
        assert L.get_tuning("c1_sweep") == 0
    finally:
        L.set_tuning_thread("c1_sweep", 0, enable=False)
        L.set_tuning("c1_sweep", 1)
    L.set_tuning("warp_tiled", -1)
    assert L.get_tuning("warp_tiled") == 1
    with pytest.raises(L.PscvError):
        L.set_tuning("no_such_knob", 1)
    with pytest.raises(L.PscvError):
        L.set_tuning_thread("no_such_knob", 1)


def test_replay_keys_follow_options_weights_and_groups():
    """Host logic of graph.replayable (the in-forward hipGraph replay): the option key sees every plain attribute of the model tree,
    a torch.distributed group attached to any sub-module disables replay, the weights key follows in-place updates and storage
    moves (`_apply`), and on CPU the decorated forward is simply the eager one."""
    from wild_deep_mvs_amd import graph as G
    from wild_deep_mvs_amd.models.MVSNet.model import MVSNet
    from wild_deep_mvs_amd.models.VisMVSNet.frontend import Frontend
    net = MVSNet("variance")
    st = {}
    k0 = G._options_key(net, st)
    assert ("", "num_depth", 192) in k0 and ("", "storage_dtype", torch.float16) in k0
    net.num_depth = 96
    assert G._options_key(net, st) != k0 and ("", "num_depth", 96) in G._options_key(net, st)
    net.depth_group = object()
    assert G._options_key(net, st) is None
    net.depth_group = None
    w0 = G._fast_weights_key(net, st)
    with torch.no_grad():
        net.cost_regularization.prob.bias.add_(1.0)
    w1 = G._fast_weights_key(net, st)
    assert w1 != w0
    net.double()                                    # storage replaced through _apply: the generation moves, the tensor list is rebuilt
    w2 = G._fast_weights_key(net, st)
    assert w2[1] == w1[1] + 1 and w2 != w1
    vis = Frontend()
    kv = G._options_key(vis, {})
    assert ("", "depth_nums", repr([32, 16, 8])) in kv
    vis.model.stage2.view_group = object()
    assert G._options_key(vis, {}) is None
    assert hasattr(MVSNet.forward, "eager")         # the decorator keeps the plain forward reachable
