"""HIP MFMA conv3d vs ATen on the CPU (bf16-rounded operands, fp32 accumulation), through the C ABI."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _util import bf16_round, check_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from wild_deep_mvs_amd import _lib as L, ops
    return L, ops


def _ref_conv(x_ncdhw, w, kind, transposed, L):
    if kind == L.CONV_T2:
        return F.conv_transpose3d(x_ncdhw, w, stride=2, padding=1, output_padding=1)
    if transposed:
        return F.conv_transpose3d(x_ncdhw, w, stride=1, padding=1)
    return F.conv3d(x_ncdhw, w, stride=1 if kind == L.CONV_S1 else 2, padding=1)


# every (c_in, c_out, kind) of the MVSNet regulariser, on sizes that are NOT multiples of the tile
MVSNET_LAYERS = [(32, 8, 0), (8, 16, 1), (16, 16, 0), (16, 32, 1), (32, 32, 0), (32, 64, 1), (64, 64, 0),
                 (64, 32, 2), (32, 16, 2), (16, 8, 2), (8, 1, 0)]


@pytest.mark.parametrize("small_tiles", [1, 0])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout,kind", MVSNET_LAYERS)
def test_conv3d_plain(env, cin, cout, kind, dtype, small_tiles):
    """small_tiles=1: 1x4x16 tiles with the output channels split over blockIdx.y (what small volumes get);
    small_tiles=0: the large-tile variant (what the full-resolution layers get)."""
    L, ops = env
    L.set_tuning("conv_small_tiles", small_tiles)
    g = torch.Generator().manual_seed(cin * 1000 + cout * 10 + kind)
    D, H, W = (6, 10, 21) if kind == L.CONV_S1 else (7, 9, 35) if kind == L.CONV_S2 else (3, 5, 19)
    B = 2
    transposed = kind == L.CONV_T2
    x = bf16_round(torch.randn(B, cin, D, H, W, generator=g))
    wshape = (cin, cout, 3, 3, 3) if transposed else (cout, cin, 3, 3, 3)
    w = bf16_round(torch.randn(wshape, generator=g) / np.sqrt(27 * cin))
    layer = ops.Conv3dLayer.build(w, kind=kind, transposed=transposed, device="cuda", dtype=dtype)
    try:
        y = ops.conv3d(ops.to_channels_last(x.cuda(), dtype), layer, out_dtype=torch.float32)   # bf16-exact values are fp16-exact
    finally:
        L.set_tuning("conv_small_tiles", 1)
    ref = _ref_conv(x, w, kind, transposed, L)
    check_close(f"conv3d {cin}->{cout} kind {kind} {dtype} operands, fp32 out", y.permute(0, 4, 1, 2, 3).cpu(), ref, max_abs=2e-3, rel_l2=1e-4)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,cout,kind,relu,with_skip", [(32, 8, 0, True, False), (64, 32, 2, True, True),
                                                           (16, 8, 2, True, True), (16, 16, 0, False, True),
                                                           (8, 16, 1, True, False)])
def test_conv3d_fused_epilogue(env, cin, cout, kind, relu, with_skip, dtype):
    """folded BN affine, ReLU before the skip add (MVSNet: skip + relu(bn(deconv))), bf16 output."""
    L, ops = env
    g = torch.Generator().manual_seed(7 + cin + cout + kind)
    D, H, W = (8, 8, 16) if kind != L.CONV_T2 else (4, 4, 8)
    transposed = kind == L.CONV_T2
    x = bf16_round(torch.randn(1, cin, D, H, W, generator=g))
    wshape = (cin, cout, 3, 3, 3) if transposed else (cout, cin, 3, 3, 3)
    w = bf16_round(torch.randn(wshape, generator=g) / np.sqrt(27 * cin))
    gamma, beta = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.3
    mean, var = torch.randn(cout, generator=g) * 0.2, torch.rand(cout, generator=g) + 0.5
    layer = ops.Conv3dLayer.build(w, kind=kind, transposed=transposed, device="cuda", bn=(gamma, beta, mean, var), relu=relu,
                                  dtype=dtype)
    ref = _ref_conv(x, w, kind, transposed, L)
    ref = F.batch_norm(ref, mean, var, gamma, beta, training=False, eps=1e-5)
    if relu:
        ref = F.relu(ref)
    skip = None
    if with_skip:
        skip = bf16_round(torch.randn(ref.shape, generator=g))
        ref = ref + skip
    y = ops.conv3d(ops.to_channels_last(x.cuda(), dtype), layer,
                   skip=None if skip is None else ops.to_channels_last(skip.cuda(), dtype))
    assert y.dtype == dtype
    ulp = 2 ** -8 if dtype == torch.bfloat16 else 2 ** -11      # half a unit in the last place, relative
    s = check_close(f"conv3d+bn{'+relu' if relu else ''}{'+skip' if with_skip else ''} {cin}->{cout} kind {kind} {dtype}",
                    y.float().permute(0, 4, 1, 2, 3).cpu(), ref, rel_l2=ulp)
    assert s["max_abs"] <= ulp * s["ref_max"] + 1e-3


def test_conv3d_channel_slices_and_relu_post(env):
    """Reads a channel slice of a wider tensor and writes a slice of a wider tensor (the Vis U-Net's
    cat([deconv, enc]) without a copy); RELU_POST = relu(bn(conv) + residual) (Vis BasicBlock)."""
    L, ops = env
    g = torch.Generator().manual_seed(11)
    x16 = bf16_round(torch.randn(1, 16, 4, 6, 16, generator=g))
    w = bf16_round(torch.randn(8, 8, 3, 3, 3, generator=g) / np.sqrt(27 * 8))
    res = bf16_round(torch.randn(1, 8, 4, 6, 16, generator=g))
    layer = ops.Conv3dLayer.build(w, kind=L.CONV_S1, device="cuda", relu=False, relu_post=True)
    out = torch.full((1, 4, 6, 16, 16), 7.0, dtype=torch.bfloat16, device="cuda")
    ops.conv3d(ops.to_channels_last(x16.cuda(), torch.bfloat16), layer, in_coff=8,
               skip=ops.to_channels_last(res.cuda(), torch.bfloat16), out=out, out_coff=8)
    ref = F.relu(F.conv3d(x16[:, 8:], w, padding=1) + res)
    check_close("slice conv + relu_post", out[..., 8:].float().permute(0, 4, 1, 2, 3).cpu(), ref, rel_l2=4e-3)
    assert float((out[..., :8].float() - 7.0).abs().max()) == 0.0, "channels outside the output slice were touched"


def test_conv3d_per_channel_floor(env):
    """floor[c] = -inf keeps a channel linear while others get ReLU in the same launch."""
    L, ops = env
    g = torch.Generator().manual_seed(5)
    x = bf16_round(torch.randn(1, 8, 4, 4, 16, generator=g))
    w = bf16_round(torch.randn(16, 8, 3, 3, 3, generator=g) / np.sqrt(27 * 8))
    floor = torch.zeros(16)
    floor[8:] = float("-inf")
    layer = ops.Conv3dLayer.build(w, kind=L.CONV_S1, device="cuda", relu=True, floor=floor)
    y = ops.conv3d(ops.to_channels_last(x.cuda(), torch.bfloat16), layer, out_dtype=torch.float32).permute(0, 4, 1, 2, 3).cpu()
    ref = F.conv3d(x, w, padding=1)
    ref = torch.cat([F.relu(ref[:, :8]), ref[:, 8:]], 1)
    check_close("per-channel floor", y, ref, max_abs=2e-3)


def test_conv3d_linearity_at_full_size(env):
    """Size-independent property at the headline volume (192x128x160, 32->8): without ReLU the layer is
    linear, conv(a) + conv(b) == conv(a + b) up to bf16 input rounding of (a + b); plus a cropped-region
    comparison against ATen around three far-apart voxels."""
    L, ops = env
    g = torch.Generator().manual_seed(3)
    D, H, W, cin, cout = 192, 128, 160, 32, 8
    a = torch.randn(1, D, H, W, cin, generator=g, dtype=torch.float32).to(torch.bfloat16)
    b = torch.randn(1, D, H, W, cin, generator=g, dtype=torch.float32).to(torch.bfloat16) * 0.5
    ab = (a.float() + b.float()).to(torch.bfloat16)
    w = bf16_round(torch.randn(cout, cin, 3, 3, 3, generator=g) / np.sqrt(27 * cin))
    layer = ops.Conv3dLayer.build(w, kind=L.CONV_S1, device="cuda")
    ya = ops.conv3d(a.cuda(), layer, out_dtype=torch.float32)
    yb = ops.conv3d(b.cuda(), layer, out_dtype=torch.float32)
    yab = ops.conv3d(ab.cuda(), layer, out_dtype=torch.float32)
    # (a+b) is rounded to bf16 once more -> tolerance of one bf16 ulp of the inputs through the contraction
    err = float((ya + yb - yab).abs().max())
    assert err <= 0.05, err
    for (d0, h0, w0) in [(0, 0, 0), (95, 63, 79), (183, 119, 151)]:
        crop = a[:, max(d0 - 1, 0):d0 + 9, max(h0 - 1, 0):h0 + 9, max(w0 - 1, 0):w0 + 9].float().permute(0, 4, 1, 2, 3)
        ref = F.conv3d(crop, w, padding=1)
        od, oh, ow = (1 if d0 else 0), (1 if h0 else 0), (1 if w0 else 0)
        ref = ref[:, :, od:od + 7, oh:oh + 7, ow:ow + 7]
        got = ya[:, d0:d0 + 7, h0:h0 + 7, w0:w0 + 7].permute(0, 4, 1, 2, 3).cpu()
        check_close(f"full-size conv crop @({d0},{h0},{w0})", got, ref, max_abs=2e-3)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape,th16", [((16, 16, 32), 1), ((16, 16, 32), 0), ((13, 11, 21), 1), ((2, 8, 16), 1), ((37, 9, 40), 1),
                                        ((9, 37, 24), 1), ((9, 37, 24), 0)])
def test_sweep_kernel_matches_brick_kernel_and_aten(env, shape, th16, dtype):
    """The depth-sweep 32->8 kernel (PSCV_CONV_S1P8) against the generic brick kernel and ATen, on sizes that are
    not multiples of the 8x16 tile / the depth chunk, odd D, with BN + ReLU + skip."""
    L, ops = env
    g = torch.Generator().manual_seed(sum(shape))
    D, H, W = shape
    x = bf16_round(torch.randn(2, 32, D, H, W, generator=g))
    w = bf16_round(torch.randn(8, 32, 3, 3, 3, generator=g) / np.sqrt(27 * 32))
    gamma, beta = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.3
    mean, var = torch.randn(8, generator=g) * 0.2, torch.rand(8, generator=g) + 0.5
    skip = bf16_round(torch.randn(2, 8, D, H, W, generator=g))
    ref = F.relu(F.batch_norm(F.conv3d(x, w, padding=1), mean, var, gamma, beta, training=False, eps=1e-5)) + skip
    xcl, scl = ops.to_channels_last(x.cuda(), dtype), ops.to_channels_last(skip.cuda(), dtype)
    outs = {}
    for use in (True, False):
        ops.USE_SWEEP_KERNEL = use
        try:
            layer = ops.Conv3dLayer.build(w, kind=L.CONV_S1, device="cuda", bn=(gamma, beta, mean, var), relu=True, dtype=dtype)
        finally:
            ops.USE_SWEEP_KERNEL = True
        assert layer.kind == (L.CONV_S1P8 if use else L.CONV_S1)
        L.set_tuning("sweep_th16", th16)       # 16-row / 512-thread tiles (when H >= 16) or 8-row / 256-thread tiles
        try:
            outs[use] = ops.conv3d(xcl, layer, skip=scl, out_dtype=torch.float32).permute(0, 4, 1, 2, 3).cpu()
        finally:
            L.set_tuning("sweep_th16", 0)
    check_close(f"sweep vs ATen {shape} {dtype}", outs[True], ref, max_abs=3e-3, rel_l2=2e-4)
    check_close(f"sweep vs brick {shape} {dtype}", outs[True], outs[False], max_abs=1e-4)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape,dc,pd,with_skip,out_f32,transposed", [((16, 16, 32), 0, 1, True, True, False), ((13, 11, 21), 5, 1, False, False, False),
                                                                     ((2, 8, 16), 0, 1, True, False, False), ((37, 9, 40), 7, 2, True, True, True),
                                                                     ((9, 37, 24), 4, 2, False, True, False), ((1, 3, 5), 0, 1, False, True, False)])
def test_kd_in_rows_sweep_matches_plane_pair_sweep_and_aten(env, shape, dc, pd, with_skip, out_f32, transposed, dtype):
    """The kd-in-rows variant of the 32 -> 8 depth sweep (pscv_set_tuning("sweep_kdm", 1): 32 x 32 x 16 MFMAs whose rows hold the
    three depth taps, one input plane per iteration, 3-slot ring) against the default plane-pair kernel and ATen: sizes that are
    not multiples of the tile, odd depth chunks (one plane of a chunk can be the volume's last), both prefetch distances, with and
    without the residual, fp32 and 16-bit outputs, a stride-1 deconv (flipped taps).  Same products, another summation order:
    fp32 outputs agree to 1e-4, not bit for bit."""
    L, ops = env
    g = torch.Generator().manual_seed(sum(shape) + dc)
    D, H, W = shape
    x = bf16_round(torch.randn(2, 32, D, H, W, generator=g))
    wshape = (32, 8, 3, 3, 3) if transposed else (8, 32, 3, 3, 3)
    w = bf16_round(torch.randn(wshape, generator=g) / np.sqrt(27 * 32))
    gamma, beta = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.3
    mean, var = torch.randn(8, generator=g) * 0.2, torch.rand(8, generator=g) + 0.5
    skip = bf16_round(torch.randn(2, 8, D, H, W, generator=g)) if with_skip else None
    ref = F.relu(F.batch_norm(_ref_conv(x, w, L.CONV_S1, transposed, L), mean, var, gamma, beta, training=False, eps=1e-5))
    if with_skip:
        ref = ref + skip
    xcl = ops.to_channels_last(x.cuda(), dtype)
    scl = ops.to_channels_last(skip.cuda(), dtype) if with_skip else None
    layer = ops.Conv3dLayer.build(w, kind=L.CONV_S1, transposed=transposed, device="cuda", bn=(gamma, beta, mean, var), relu=True, dtype=dtype)
    assert layer.kind == L.CONV_S1P8
    outs = {}
    for kdm in (0, 1, 2):
        L.set_tuning("sweep_kdm", kdm); L.set_tuning("sweep_dc", dc if kdm else 0); L.set_tuning("sweep_kdm_pd", pd)
        try:
            out = torch.full((2, D, H, W, 8), float("nan"), dtype=torch.float32 if out_f32 else dtype, device="cuda")
            ops.conv3d(xcl, layer, skip=scl, out=out)
            outs[kdm] = out.float().permute(0, 4, 1, 2, 3).cpu()
        finally:
            L.set_tuning("sweep_kdm", 0); L.set_tuning("sweep_dc", 0); L.set_tuning("sweep_kdm_pd", 0)
    ulp = 2 ** -8 if dtype == torch.bfloat16 else 2 ** -11
    for kdm in (1, 2):
        check_close(f"kd-in-rows({kdm}) vs ATen {shape} {dtype}", outs[kdm], ref, max_abs=3e-3 if out_f32 else None, rel_l2=2e-4 if out_f32 else 2 * ulp)
        if out_f32:
            check_close(f"kd-in-rows({kdm}) vs plane-pair {shape} {dtype}", outs[kdm], outs[0], max_abs=1e-4)
        else:   # 16-bit stores: the two summation orders may round a value to neighbouring representable numbers
            assert float((outs[kdm] - outs[0]).abs().max()) <= 2 * ulp * float(outs[0].abs().max()) + 1e-6
    # NaN in, NaN out (relu_floor keeps it) -- and only where the receptive field holds it
    xn = xcl.clone()
    xn[0, D // 2, H // 2, W // 2, 3] = float("nan")
    L.set_tuning("sweep_kdm", 1)
    try:
        yn = ops.conv3d(xn, layer, skip=scl, out_dtype=torch.float32)
    finally:
        L.set_tuning("sweep_kdm", 0)
    nan = torch.isnan(yn[0]).any(-1)
    d0, h0, w0 = D // 2, H // 2, W // 2
    exp = torch.zeros_like(nan)
    exp[max(d0 - 1, 0):d0 + 2, max(h0 - 1, 0):h0 + 2, max(w0 - 1, 0):w0 + 2] = True
    assert bool((nan == exp).all()) and not bool(torch.isnan(yn[1]).any())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("cin,shape,transposed,dc", [(8, (16, 16, 32), False, 0), (8, (13, 11, 21), False, 4), (8, (37, 9, 40), True, 0),
                                                     (16, (16, 16, 32), False, 0), (16, (9, 37, 24), False, 6), (16, (2, 8, 16), True, 0),
                                                     (8, (2, 3, 5), False, 0)])
def test_narrow_sweep_kernel_matches_brick_kernel_and_aten(env, cin, shape, transposed, dc, dtype):
    """The 8|16 -> 8 depth-sweep kernel (PSCV_CONV_S1P8, the Vis U-Net's full-resolution layers) against the generic brick kernel
    and ATen: sizes off the 8x16 tile, odd D, forced depth-chunk seams, BN + skip + post-ReLU, input / skip / output channel
    slices of wider tensors (the U-Net's concat buffer), and stride-1 ConvTranspose3d weights (the training path's adjoints)."""
    L, ops = env
    g = torch.Generator().manual_seed(cin + sum(shape))
    D, H, W = shape
    wide = bf16_round(torch.randn(2, 24, D, H, W, generator=g))               # the conv reads channels [8, 8 + cin)
    x = wide[:, 8:8 + cin]
    if transposed:
        w = bf16_round(torch.randn(cin, 8, 3, 3, 3, generator=g) / np.sqrt(27 * cin))
        conv = F.conv_transpose3d(x, w, padding=1)
    else:
        w = bf16_round(torch.randn(8, cin, 3, 3, 3, generator=g) / np.sqrt(27 * cin))
        conv = F.conv3d(x, w, padding=1)
    gamma, beta = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.3
    mean, var = torch.randn(8, generator=g) * 0.2, torch.rand(8, generator=g) + 0.5
    skip_wide = bf16_round(torch.randn(2, 16, D, H, W, generator=g))          # skip = channels [4, 12)
    ref = F.relu(F.batch_norm(conv, mean, var, gamma, beta, training=False, eps=1e-5) + skip_wide[:, 4:12])
    xcl, scl = ops.to_channels_last(wide.cuda(), dtype), ops.to_channels_last(skip_wide.cuda(), dtype)
    outs = {}
    for use in (True, False):
        ops.USE_SWEEP_KERNEL = use
        try:
            layer = ops.Conv3dLayer.build(w, kind=L.CONV_S1, transposed=transposed, device="cuda", bn=(gamma, beta, mean, var),
                                          relu_post=True, dtype=dtype)
        finally:
            ops.USE_SWEEP_KERNEL = True
        assert layer.kind == (L.CONV_S1P8 if use else L.CONV_S1)
        out = torch.full((2, D, H, W, 16), 7.0, dtype=dtype, device="cuda")     # the conv writes channels [8, 16)
        L.set_tuning("sweep_dc", dc)
        try:
            ops.conv3d(xcl, layer, in_coff=8, skip=scl, skip_coff=4, out=out, out_coff=8)
        finally:
            L.set_tuning("sweep_dc", 0)
        assert bool((out[..., :8] == 7.0).all())
        outs[use] = out[..., 8:].float().permute(0, 4, 1, 2, 3).cpu()
    ulp = 2 ** -8 if dtype == torch.bfloat16 else 2 ** -11
    check_close(f"narrow sweep vs ATen cin={cin} {shape} {dtype}", outs[True], ref, max_abs=ulp * float(ref.abs().max()) + 2e-3)
    check_close(f"narrow sweep vs brick cin={cin} {shape} {dtype}", outs[True], outs[False], max_abs=ulp * float(ref.abs().max()) + 1e-4)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape,transposed,dc", [((16, 16, 32), False, 0), ((13, 11, 21), False, 4), ((37, 9, 40), True, 0), ((2, 3, 5), False, 0),
                                                 ((9, 37, 24), False, 6)])
def test_sweep16_kernel_matches_brick_kernel_and_aten(env, shape, transposed, dc, dtype):
    """The 16 -> 16 depth-sweep variant (CVP-MVSNet's full-resolution conv0 / conv0a, MVSNet's conv2) against the brick kernel and
    ATen: off-tile sizes, odd D, forced chunk seams, BN + ReLU + skip, channel slices, stride-1 ConvTranspose3d weights."""
    L, ops = env
    g = torch.Generator().manual_seed(16 + sum(shape))
    D, H, W = shape
    wide = bf16_round(torch.randn(2, 32, D, H, W, generator=g))
    x = wide[:, 8:24]
    if transposed:
        w = bf16_round(torch.randn(16, 16, 3, 3, 3, generator=g) / np.sqrt(27 * 16))
        conv = F.conv_transpose3d(x, w, padding=1)
    else:
        w = bf16_round(torch.randn(16, 16, 3, 3, 3, generator=g) / np.sqrt(27 * 16))
        conv = F.conv3d(x, w, padding=1)
    gamma, beta = torch.rand(16, generator=g) + 0.5, torch.randn(16, generator=g) * 0.3
    mean, var = torch.randn(16, generator=g) * 0.2, torch.rand(16, generator=g) + 0.5
    skip_wide = bf16_round(torch.randn(2, 24, D, H, W, generator=g))
    ref = F.relu(F.batch_norm(conv, mean, var, gamma, beta, training=False, eps=1e-5)) + skip_wide[:, 4:20]
    xcl, scl = ops.to_channels_last(wide.cuda(), dtype), ops.to_channels_last(skip_wide.cuda(), dtype)
    outs = {}
    for use in (True, False):
        ops.USE_SWEEP_KERNEL, ops.SWEEP16 = use, True      # (the brick kernel is the default for 16 -> 16; SWEEP16 selects the sweep)
        try:
            layer = ops.Conv3dLayer.build(w, kind=L.CONV_S1, transposed=transposed, device="cuda", bn=(gamma, beta, mean, var), relu=True,
                                          dtype=dtype)
        finally:
            ops.USE_SWEEP_KERNEL, ops.SWEEP16 = True, False
        assert layer.kind == (L.CONV_S1P8 if use else L.CONV_S1)
        out = torch.full((2, D, H, W, 24), 7.0, dtype=dtype, device="cuda")
        L.set_tuning("sweep_dc", dc)
        try:
            ops.conv3d(xcl, layer, in_coff=8, skip=scl, skip_coff=4, out=out, out_coff=4)
        finally:
            L.set_tuning("sweep_dc", 0)
        assert bool((out[..., :4] == 7.0).all()) and bool((out[..., 20:] == 7.0).all())
        outs[use] = out[..., 4:20].float().permute(0, 4, 1, 2, 3).cpu()
    ulp = 2 ** -8 if dtype == torch.bfloat16 else 2 ** -11
    check_close(f"sweep16 vs ATen {shape} {dtype}", outs[True], ref, max_abs=ulp * float(ref.abs().max()) + 2e-3)
    check_close(f"sweep16 vs brick {shape} {dtype}", outs[True], outs[False], max_abs=ulp * float(ref.abs().max()) + 1e-4)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("cin,shape,out_dtype", [(8, (16, 16, 32), torch.float32), (8, (13, 11, 45), torch.float32),
                                                 (16, (9, 8, 33), torch.float32), (8, (5, 17, 64), None)])
def test_one_channel_dot2_kernel_matches_mfma_kernel_and_aten(env, cin, shape, out_dtype, dtype):
    """The vector-ALU dot2 sweep kernel for the 1-channel heads (PSCV_CONV_S1C1) against the MFMA brick kernel and
    ATen: bias, sizes off the 8x32 tile and off the depth chunk, fp32 and 16-bit outputs."""
    L, ops = env
    g = torch.Generator().manual_seed(cin + sum(shape))
    D, H, W = shape
    x = bf16_round(torch.randn(2, cin, D, H, W, generator=g))
    w = bf16_round(torch.randn(1, cin, 3, 3, 3, generator=g) / np.sqrt(27 * cin))
    bias = torch.randn(1, generator=g)
    ref = F.conv3d(x, w, bias, padding=1)
    xcl = ops.to_channels_last(x.cuda(), dtype)
    outs = {}
    for use in (True, False):
        ops.USE_SWEEP_KERNEL = use
        try:
            layer = ops.Conv3dLayer.build(w, kind=L.CONV_S1, device="cuda", conv_bias=bias, dtype=dtype)
        finally:
            ops.USE_SWEEP_KERNEL = True
        assert layer.kind == (L.CONV_S1C1 if use else L.CONV_S1)
        y = ops.conv3d(xcl, layer, out_dtype=out_dtype)
        assert y.dtype == (out_dtype or dtype)
        outs[use] = y.float().permute(0, 4, 1, 2, 3).cpu()
    tol = 3e-3 if out_dtype is not None else (2 ** -8 if dtype == torch.bfloat16 else 2 ** -11) * float(ref.abs().max()) + 1e-3
    check_close(f"dot2 kernel vs ATen cin={cin} {shape} {dtype}", outs[True], ref, max_abs=tol)
    check_close(f"dot2 kernel vs MFMA kernel cin={cin} {shape} {dtype}", outs[True], outs[False], max_abs=tol)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(4, 8, 16), (3, 5, 19), (7, 9, 33)])
def test_parity_pair_deconv_matches_generic_kernel_and_aten(env, shape, dtype):
    """The 16->8 stride-2 transposed conv on the parity-pair kernel (PSCV_CONV_T2P8) against the generic T2 kernel and
    ATen, with BN + ReLU + skip, sizes off the 2x4x16 tile, and a channel-slice output (the Vis U-Net's cat buffer)."""
    L, ops = env
    g = torch.Generator().manual_seed(sum(shape))
    D, H, W = shape
    x = bf16_round(torch.randn(2, 16, D, H, W, generator=g))
    w = bf16_round(torch.randn(16, 8, 3, 3, 3, generator=g) / np.sqrt(27 * 16))
    gamma, beta = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.3
    mean, var = torch.randn(8, generator=g) * 0.2, torch.rand(8, generator=g) + 0.5
    skip = bf16_round(torch.randn(2, 8, 2 * D, 2 * H, 2 * W, generator=g))
    ref = F.relu(F.batch_norm(F.conv_transpose3d(x, w, stride=2, padding=1, output_padding=1), mean, var, gamma, beta,
                              training=False, eps=1e-5)) + skip
    xcl, scl = ops.to_channels_last(x.cuda(), dtype), ops.to_channels_last(skip.cuda(), dtype)
    outs = {}
    for use in (True, False):
        ops.USE_SWEEP_KERNEL = use
        try:
            layer = ops.Conv3dLayer.build(w, kind=L.CONV_T2, transposed=True, device="cuda", bn=(gamma, beta, mean, var),
                                          relu=True, dtype=dtype)
        finally:
            ops.USE_SWEEP_KERNEL = True
        assert layer.kind == (L.CONV_T2P8 if use else L.CONV_T2)
        outs[use] = ops.conv3d(xcl, layer, skip=scl, out_dtype=torch.float32).permute(0, 4, 1, 2, 3).cpu()
    check_close(f"t2p8 vs ATen {shape} {dtype}", outs[True], ref, max_abs=3e-3, rel_l2=2e-4)
    check_close(f"t2p8 vs generic T2 {shape} {dtype}", outs[True], outs[False], max_abs=1e-4)
    # write into channels [0, 8) of a 16-channel buffer, linear epilogue (Vis decoder)
    lin = ops.Conv3dLayer.build(w, kind=L.CONV_T2, transposed=True, device="cuda", dtype=dtype)
    buf = torch.full((2, 2 * D, 2 * H, 2 * W, 16), 3.0, dtype=dtype, device="cuda")
    ops.conv3d(xcl, lin, out=buf, out_coff=0)
    want = F.conv_transpose3d(x, w, stride=2, padding=1, output_padding=1)
    check_close("t2p8 slice output", buf[..., :8].float().permute(0, 4, 1, 2, 3).cpu(), want,
                max_abs=(2 ** -8 if dtype == torch.bfloat16 else 2 ** -11) * float(want.abs().max()) + 1e-3)
    assert float((buf[..., 8:].float() - 3.0).abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_device_weight_packing_equals_host_packing(dtype):
    """pscv_pack_conv3d_weights_device (one launch, used when the weights live on the GPU: every training step) writes
    exactly the bits of the host packer for every layout (dense S1 / S2 / T2 incl. flipped-tap transposed S1, and the
    three special kernels' layouts)."""
    from wild_deep_mvs_amd import _lib as L, ops
    g = torch.Generator().manual_seed(5)
    cases = [(8, 32, L.CONV_S1, False), (32, 8, L.CONV_S1, True), (64, 32, L.CONV_S1, True), (16, 8, L.CONV_S2, False),
             (64, 32, L.CONV_S2, False), (64, 32, L.CONV_T2, True), (32, 16, L.CONV_T2, True), (8, 32, L.CONV_S1P8, False), (8, 8, L.CONV_S1P8, False), (8, 16, L.CONV_S1P8, False), (8, 8, L.CONV_S1P8, True), (16, 16, L.CONV_S1P8, False), (16, 16, L.CONV_S1P8, True),
             (16, 8, L.CONV_T2P8, True), (1, 8, L.CONV_S1C1, False), (1, 16, L.CONV_S1C1, False)]
    for a, b, kind, tr in cases:
        w = torch.randn(a, b, 3, 3, 3, generator=g)
        host = torch.from_numpy(ops.pack_conv3d_weights(w, kind, tr, dtype).view(np.int16))
        dev = ops.pack_conv3d_weights_device(w.cuda(), kind, tr, dtype).cpu()
        assert torch.equal(host, dev), (a, b, kind, tr)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("cout,shape,dc_slots,out_f32", [(16, (16, 16, 32), 0, False), (16, (13, 11, 21), 0, True), (32, (9, 37, 40), 0, False),
                                                          (16, (37, 9, 70), 4, False), (24, (7, 33, 35), 0, False), (32, (2, 3, 5), 0, False),
                                                          (16, (30, 20, 34), 6, False)])
def test_stride2_sweep_kernel_matches_brick_kernel_and_aten(env, cout, shape, dc_slots, out_f32, dtype):
    """The stride-2 depth-sweep kernel (csrc/conv3d_sweep_s2.hip: MVSNet's conv1 8 -> 16, the Vis U-Net's fused strided conv 8 -> 32)
    against the brick kernel (same packed weights) and ATen: odd sizes in every dimension (output = ceil(size / 2)), forced chunk seams
    ("s2s_slots" sizes the depth chunks), BN + ReLU + skip, channel slices in and out, a ragged last N-tile (24 channels), fp32 out."""
    L, ops = env
    g = torch.Generator().manual_seed(80 + cout + sum(shape))
    D, H, W = shape
    wide = bf16_round(torch.randn(2, 24, D, H, W, generator=g))
    x = wide[:, 8:16]
    w = bf16_round(torch.randn(cout, 8, 3, 3, 3, generator=g) / np.sqrt(27 * 8))
    conv = F.conv3d(x, w, stride=2, padding=1)
    Do, Ho, Wo = conv.shape[2:]
    gamma, beta = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.3
    mean, var = torch.randn(cout, generator=g) * 0.2, torch.rand(cout, generator=g) + 0.5
    skip_wide = bf16_round(torch.randn(2, cout + 8, Do, Ho, Wo, generator=g))
    ref = F.relu(F.batch_norm(conv, mean, var, gamma, beta, training=False, eps=1e-5)) + skip_wide[:, 4:4 + cout]
    xcl, scl = ops.to_channels_last(wide.cuda(), dtype), ops.to_channels_last(skip_wide.cuda(), dtype)
    layer = ops.Conv3dLayer.build(w, kind=L.CONV_S2, device="cuda", bn=(gamma, beta, mean, var), relu=True, dtype=dtype)
    odt = torch.float32 if out_f32 else dtype
    outs = {}
    for sweep in (2, 0):                 # 2: the sweep at any size; 0: the brick kernel
        out = torch.full((2, Do, Ho, Wo, cout + 8), 7.0, dtype=odt, device="cuda")
        L.set_tuning("conv_s2_sweep", sweep); L.set_tuning("s2s_slots", dc_slots * 2 * ((Ho + 7) // 8) * ((Wo + 15) // 16))
        try:
            ops.conv3d(xcl, layer, in_coff=8, skip=scl, skip_coff=4, out=out, out_coff=4, out_dtype=odt)
        finally:
            L.set_tuning("conv_s2_sweep", 1); L.set_tuning("s2s_slots", 0)
        assert bool((out[..., :4] == 7.0).all()) and bool((out[..., 4 + cout:] == 7.0).all())
        outs[sweep] = out[..., 4:4 + cout].float().permute(0, 4, 1, 2, 3).cpu()
    ulp = 0.0 if out_f32 else (2 ** -8 if dtype == torch.bfloat16 else 2 ** -11)
    check_close(f"s2 sweep vs ATen {shape} -> {cout} {dtype}", outs[2], ref, max_abs=ulp * float(ref.abs().max()) + 2e-3)
    check_close(f"s2 sweep vs brick {shape} -> {cout} {dtype}", outs[2], outs[0], max_abs=ulp * float(ref.abs().max()) + 1e-4)
    # the models' call: no skip tensor, 16-bit output (the kernel's straight-line epilogue instantiation)
    plain = {}
    ref_plain = F.relu(F.batch_norm(conv, mean, var, gamma, beta, training=False, eps=1e-5))
    for sweep in (2, 0):
        L.set_tuning("conv_s2_sweep", sweep)
        try:
            plain[sweep] = ops.conv3d(xcl, layer, in_coff=8).float().permute(0, 4, 1, 2, 3).cpu()
        finally:
            L.set_tuning("conv_s2_sweep", 1)
    u16 = 2 ** -8 if dtype == torch.bfloat16 else 2 ** -11
    check_close(f"s2 sweep (plain) vs ATen {shape} -> {cout} {dtype}", plain[2], ref_plain, max_abs=u16 * float(ref_plain.abs().max()) + 2e-3)
    check_close(f"s2 sweep (plain) vs brick {shape} -> {cout} {dtype}", plain[2], plain[0], max_abs=u16 * float(ref_plain.abs().max()) + 1e-4)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape,out_dtype,with_skip", [((16, 16, 32), torch.float32, False), ((13, 11, 45), torch.float32, False),
                                                       ((37, 9, 40), torch.float32, False), ((5, 17, 64), None, False),
                                                       ((48, 6, 70), torch.float32, True), ((25, 12, 33), None, True)])
def test_one_channel_head_depth_sweep_matches_brick_variant_and_aten(env, shape, out_dtype, with_skip, dtype):
    """The depth-sweep variant of the 1-channel head kernel (csrc/conv3d_c1.hip: 16-slot plane ring, planes requested one 6-plane block
    ahead with raw buffer loads; what MVSNet's `prob` runs at D = 192) against the brick variant (pscv_set_tuning("c1_sweep", 0)) and
    ATen: depths that are not multiples of 6, tiles off the 4 x 32 grid, a channel-slice input, bias, 16-bit and fp32 outputs, skip."""
    L, ops = env
    g = torch.Generator().manual_seed(31 + sum(shape))
    D, H, W = shape
    wide = bf16_round(torch.randn(2, 16, D, H, W, generator=g))
    x = wide[:, 8:16]
    w = bf16_round(torch.randn(1, 8, 3, 3, 3, generator=g) / np.sqrt(27 * 8))
    bias = torch.randn(1, generator=g)
    skip = bf16_round(torch.randn(2, 1, D, H, W, generator=g)) if with_skip else None
    ref = F.conv3d(x, w, bias, padding=1) + (skip if with_skip else 0.0)
    xcl = ops.to_channels_last(wide.cuda(), dtype)
    scl = ops.to_channels_last(skip.cuda(), dtype) if with_skip else None
    layer = ops.Conv3dLayer.build(w, kind=L.CONV_S1, device="cuda", conv_bias=bias, dtype=dtype)
    assert layer.kind == L.CONV_S1C1
    outs = {}
    for sweep in (2, 0):                # 2: the sweep at any depth; 0: the brick variant
        L.set_tuning("c1_sweep", sweep)
        try:
            y = ops.conv3d(xcl, layer, in_coff=8, skip=scl, out_dtype=out_dtype)
        finally:
            L.set_tuning("c1_sweep", 1)
        outs[sweep] = y.float().permute(0, 4, 1, 2, 3).cpu()
    tol = 3e-3 if out_dtype is not None else (2 ** -8 if dtype == torch.bfloat16 else 2 ** -11) * float(ref.abs().max()) + 1e-3
    check_close(f"c1 sweep vs ATen {shape} {dtype}", outs[2], ref, max_abs=tol)
    check_close(f"c1 sweep vs brick variant {shape} {dtype}", outs[2], outs[0], max_abs=tol if out_dtype is None else 1e-5)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(24, 8, 32), (37, 9, 40), (48, 6, 70), (192, 16, 32)])
def test_fused_prob_softargmin_tail_equals_separate_launches(env, shape, dtype):
    """pscv_prob_softargmin (the 1-channel head's depth sweep emits per-chunk softmax partials, one merge launch gives depth and the
    4-plane confidence) against pscv_conv3d + pscv_softargmin on the same input: identical logits, depth within 2e-6 of the depth
    range, confidence within 2e-5; depths that are not multiples of 6, ragged tiles, several chunks, batch of two with different
    depth planes.  Reference semantics: models/MVSNet/model.py:72,82,207-215."""
    L, ops = env
    g = torch.Generator().manual_seed(61 + sum(shape))
    D, H, W = shape
    x = bf16_round(torch.randn(2, 8, D, H, W, generator=g))
    w = bf16_round(torch.randn(1, 8, 3, 3, 3, generator=g) * 1.5 / np.sqrt(27 * 8))     # (logit spread of a few units: a peaked softmax)
    bias = torch.randn(1, generator=g)
    dv = torch.stack([torch.linspace(2.0, 6.0, D), torch.linspace(1.0, 9.0, D)]).cuda().contiguous()
    xcl = ops.to_channels_last(x.cuda(), dtype)
    layer = ops.Conv3dLayer.build(w, kind=L.CONV_S1, device="cuda", conv_bias=bias, dtype=dtype)
    L.set_tuning("c1_sweep", 2)
    try:
        fused = ops.prob_softargmin(xcl, layer, dv)
        logits = ops.conv3d(xcl, layer, out_dtype=torch.float32).view(2, D, H, W)
    finally:
        L.set_tuning("c1_sweep", 1)
    assert fused is not None
    sep = ops.softargmin(logits, dv, want_conf=True, conf_mode=0)
    assert torch.equal(fused["logits"], logits)
    check_close(f"fused tail depth {shape} {dtype}", fused["depth"].cpu(), sep["depth"].cpu(), max_abs=2e-6 * 9.0)
    check_close(f"fused tail confidence {shape} {dtype}", fused["conf"].cpu(), sep["conf"].cpu(), max_abs=2e-5)
    # and against the definition
    p = torch.softmax(logits.double(), 1)
    check_close(f"fused tail depth vs softmax {shape}", fused["depth"].cpu(), (p * dv.double().view(2, D, 1, 1)).sum(1).float().cpu(), max_abs=2e-5 * 9.0)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("with_skip", [True, False])
@pytest.mark.parametrize("shape", [(12, 8, 14), (9, 4, 14), (24, 8, 28), (7, 5, 9), (13, 12, 30), (96, 16, 20)])
def test_tail_sweep_equals_the_two_layers(env, shape, with_skip, dtype):
    """pscv_tail_sweep (round 5: conv11^T + BatchNorm + ReLU + skip add and the prob head as ONE depth sweep; the 8-channel
    full-resolution volume lives in an LDS plane ring) against pscv_conv3d(T2P8) followed by pscv_conv3d(S1C1, depth-sweep variant)
    on the same inputs: IDENTICAL logits (same MFMA chains, same epilogue operation order, same 16-bit rounding of the intermediate),
    on input shapes (Di, Hi, Wi) that are / are not multiples of the tile (8 x 28 output pixels = 4 x 14 input voxels), of the
    6-plane block and of the depth chunk, several chunks, batch of two, folded affine + ReLU + per-channel floor.  And against ATen.
    Reference semantics: models/MVSNet/model.py:67-72,81-82."""
    L, ops = env
    g = torch.Generator().manual_seed(17 + sum(shape) + int(with_skip))
    Di, Hi, Wi = shape
    B = 2
    x = bf16_round(torch.randn(B, 16, Di, Hi, Wi, generator=g))
    wu = bf16_round(torch.randn(16, 8, 3, 3, 3, generator=g) / np.sqrt(27 * 16 / 8))
    wh = bf16_round(torch.randn(1, 8, 3, 3, 3, generator=g) / np.sqrt(27 * 8))
    bn = (torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.1, torch.randn(8, generator=g) * 0.1, torch.rand(8, generator=g) + 0.5)
    bias = torch.randn(1, generator=g)
    skip = bf16_round(torch.randn(B, 8, 2 * Di, 2 * Hi, 2 * Wi, generator=g)) if with_skip else None
    up = ops.Conv3dLayer.build(wu, kind=L.CONV_T2, transposed=True, device="cuda", bn=bn, relu=True, dtype=dtype)
    head = ops.Conv3dLayer.build(wh, kind=L.CONV_S1, device="cuda", conv_bias=bias, dtype=dtype)
    assert up.kind == L.CONV_T2P8 and head.kind == L.CONV_S1C1
    xcl = ops.to_channels_last(x.cuda(), dtype)
    scl = ops.to_channels_last(skip.cuda(), dtype) if with_skip else None
    outs = {}
    for nbk in (0, 2, 1):
        L.set_tuning("tail_nbk", nbk)
        try:
            outs[nbk] = ops.tail_sweep(xcl, up, head, skip=scl)
        finally:
            L.set_tuning("tail_nbk", 0)
        assert outs[nbk] is not None
    u11 = ops.conv3d(xcl, up, skip=scl)
    L.set_tuning("c1_sweep", 2)              # the depth-sweep variant of the head: the accumulation chains the fused kernel uses
    try:
        two = ops.conv3d(u11, head, out_dtype=torch.float32).view(B, 2 * Di, 2 * Hi, 2 * Wi)
    finally:
        L.set_tuning("c1_sweep", 1)
    default = ops.conv3d(u11, head, out_dtype=torch.float32).view(B, 2 * Di, 2 * Hi, 2 * Wi)
    torch.cuda.synchronize()
    for nbk, o in outs.items():
        assert torch.isfinite(o).all()
        ne = int((o != two).sum())
        assert ne == 0, f"tail_nbk={nbk}: {ne} of {o.numel()} logits differ from the two launches (max {float((o - two).abs().max()):.3e})"
    check_close(f"tail sweep vs default head variant {shape}", outs[0].cpu(), default.cpu(), max_abs=1e-4)
    # the regression folded into the sweep (per-chunk softmax statistics + one merge launch) against pscv_softargmin on the same logits:
    # depth within 2e-6 of the depth range, confidence within 2e-5; batch items with different depth planes; several chunk counts
    D2 = 2 * Di
    dv = torch.stack([torch.linspace(2.0, 6.0, D2), torch.linspace(1.0, 9.0, D2)]).cuda().contiguous()
    sep = ops.softargmin(two.contiguous(), dv, want_conf=True, conf_mode=0)
    for nbk in (0, 2, 3):
        L.set_tuning("tail_nbk", nbk)
        try:
            fz = ops.tail_sweep(xcl, up, head, skip=scl, regress=dv)
        finally:
            L.set_tuning("tail_nbk", 0)
        assert isinstance(fz, dict) and torch.equal(fz["logits"], two)
        check_close(f"tail sweep regression depth {shape} nbk={nbk}", fz["depth"].cpu(), sep["depth"].cpu(), max_abs=2e-6 * 9.0)
        check_close(f"tail sweep regression confidence {shape} nbk={nbk}", fz["conf"].cpu(), sep["conf"].cpu(), max_abs=2e-5)
    # against ATen on the 16-bit-rounded operands (the intermediate rounded like the engine stores it)
    scale = bn[0] / torch.sqrt(bn[3] + 1e-5)
    y = F.conv_transpose3d(x, wu, stride=2, padding=1, output_padding=1) * scale.view(1, 8, 1, 1, 1) + (bn[1] - bn[2] * scale).view(1, 8, 1, 1, 1)
    y = torch.relu(y) + (skip if with_skip else 0)
    y = y.to(dtype).float()
    ref = F.conv3d(y, wh, padding=1) + bias.view(1, 1, 1, 1, 1)
    check_close(f"tail sweep vs ATen {shape} {dtype}", outs[0].cpu(), ref[:, 0], max_abs=3e-2 if dtype == torch.bfloat16 else 6e-3, rel_l2=4e-3 if dtype == torch.bfloat16 else 6e-4)


@pytest.mark.parametrize("cin,cout,kind,shape", [(32, 8, 0, (16, 16, 32)),      # conv0's depth sweep
                                                  (8, 8, 0, (16, 16, 32)),       # narrow sweep
                                                  (8, 16, 1, (16, 16, 32)),      # stride-2 (brick at this size; sweep forced below)
                                                  (16, 16, 0, (6, 10, 21)),      # brick kernel
                                                  (16, 8, 2, (4, 8, 16)),        # parity-pair deconv
                                                  (8, 1, 0, (16, 16, 32))])      # one-channel head
@pytest.mark.parametrize("relu", [True, False])
def test_conv3d_epilogue_propagates_nan_like_torch_relu(env, cin, cout, kind, shape, relu):
    """A NaN activation must stay NaN through `[relu](scale * conv + bias)` exactly where ATen's conv + F.relu give NaN
    (round-2 advisor finding: max(NaN, -inf) = -inf / max(NaN, 0) = 0 hid divergence from isfinite checks)."""
    L, ops = env
    g = torch.Generator().manual_seed(3)
    D, H, W = shape
    transposed = kind == L.CONV_T2
    x = bf16_round(torch.randn(1, cin, D, H, W, generator=g))
    x[0, 1, D // 2, H // 2, W // 2] = float("nan")
    wshape = (cin, cout, 3, 3, 3) if transposed else (cout, cin, 3, 3, 3)
    w = bf16_round(torch.randn(wshape, generator=g) / np.sqrt(27 * cin))
    layer = ops.Conv3dLayer.build(w, kind=kind, transposed=transposed, device="cuda", relu=relu, dtype=torch.float16)
    L.set_tuning("conv_s2_sweep", 2)
    try:
        y = ops.conv3d(ops.to_channels_last(x.cuda(), torch.float16), layer, out_dtype=torch.float32)
    finally:
        L.set_tuning("conv_s2_sweep", 1)
    ref = _ref_conv(x, w, kind, transposed, L)
    if relu:
        ref = F.relu(ref)
    got = y.permute(0, 4, 1, 2, 3).cpu()
    assert torch.isnan(ref).any()
    # every voxel ATen poisons is NaN here too (none became -inf / 0); kernels that pad their reduction with ZERO weights (the
    # plane-pair rows of the depth sweeps, the banded depth-in-rows operand of the 1-channel head) also poison the planes whose
    # zero-weight rows touch the NaN (0 * NaN = NaN inside the MFMA): at most as many again, all within 2 planes of the NaN
    gn, rn = torch.isnan(got), torch.isnan(ref)
    assert bool((gn | ~rn).all()), (int(gn.sum()), int(rn.sum()))
    assert int(gn.sum()) <= 2 * int(rn.sum())
    dpl = torch.nonzero(gn)[:, 2]
    scale = 2 if kind == L.CONV_T2 else 1
    assert int((dpl.float() / scale - (D // 2) / (2 if kind == L.CONV_S2 else 1)).abs().max()) <= 3
    assert not torch.isinf(got).any()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("cout", [8, 16])
@pytest.mark.parametrize("shape", [(16, 16, 32), (13, 11, 21), (2, 3, 5), (40, 9, 37)])
def test_two_tensor_input_equals_the_concatenated_input(env, shape, cout, dtype):
    """pscv_conv3d_cat2: the 16 input channels gathered from two tensors (dense, and 8-channel slices of wider ones) give the
    bits the same kernel gives on torch.cat([a, b], channel) -- the Vis U-Net's decoder conv (nn_utils.py:269-272) without
    the concatenated buffer."""
    L, ops = env
    D, H, W = shape
    g = torch.Generator().manual_seed(D * 100 + W + cout)
    B = 2
    a = bf16_round(torch.randn(B, D, H, W, 8, generator=g)).cuda().to(dtype)
    b = bf16_round(torch.randn(B, D, H, W, 8, generator=g)).cuda().to(dtype)
    sk = bf16_round(torch.randn(B, D, H, W, cout, generator=g)).cuda().to(dtype)
    w = bf16_round(torch.randn(cout, 16, 3, 3, 3, generator=g) / np.sqrt(27 * 16))
    bn = tuple(torch.rand(cout, generator=g) + 0.5 for _ in range(2)) + (torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5)
    layer = ops.Conv3dLayer.build(w, kind=L.CONV_S1P8, device="cuda", dtype=dtype, bn=bn, relu=True)
    cat = torch.cat([a, b], dim=4).contiguous()
    want = ops.conv3d(cat, layer, skip=sk)
    got = ops.conv3d(a, layer, x2=b, skip=sk)
    assert torch.equal(got, want)
    # slices of wider tensors on both sides, output into a channel slice, fp32 output
    wa = torch.full((B, D, H, W, 24), float("nan"), dtype=dtype, device="cuda")
    wb = torch.full((B, D, H, W, 16), float("nan"), dtype=dtype, device="cuda")
    wa[..., 16:24] = a
    wb[..., 0:8] = b
    out = torch.full((B, D, H, W, 2 * cout), -7.0, dtype=torch.float32, device="cuda")
    ops.conv3d(wa, layer, in_coff=16, x2=wb, x2_coff=0, out=out, out_coff=cout)
    want32 = ops.conv3d(cat, layer, out_dtype=torch.float32)
    assert torch.equal(out[..., cout:], want32) and bool((out[..., :cout] == -7.0).all())
    # and against ATen
    ref = F.relu(F.batch_norm(F.conv3d(cat.float().permute(0, 4, 1, 2, 3).cpu(), w, padding=1), bn[2], bn[3], bn[0], bn[1], eps=1e-5))
    check_close(f"cat2 16->{cout} {dtype}", want32.permute(0, 4, 1, 2, 3).cpu(), ref, max_abs=3e-3, rel_l2=2e-4)


def test_two_tensor_input_rejects_other_layers(env):
    L, ops = env
    x = torch.zeros(1, 4, 4, 8, 8, dtype=torch.float16, device="cuda")
    lay = ops.Conv3dLayer.build(torch.zeros(16, 16, 3, 3, 3), kind=L.CONV_S2, device="cuda", dtype=torch.float16)
    with pytest.raises(ValueError):
        ops.conv3d(x, lay, x2=x)
    lay = ops.Conv3dLayer.build(torch.zeros(8, 16, 3, 3, 3), kind=L.CONV_S1P8, device="cuda", dtype=torch.float16)
    with pytest.raises(ValueError):
        ops.conv3d(x, lay, x2=x[:, :2])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cout,shape", [(64, (6, 10, 21)), (32, (9, 17, 40)), (64, (4, 8, 16))])
def test_tall_64_channel_tiles_equal_the_4x4_tiles(env, cout, shape, dtype):
    """64-channel stride-1 layers on 4x8x16 tiles (conv_tall64 = 2: at any size; CVP's 64 -> 64 / 64 -> 32 at full size get them
    by default) against the 4x4x16 tiles (0): same reduction order per output, same bits; with BN + ReLU + skip, ragged sizes."""
    L, ops = env
    D, H, W = shape
    g = torch.Generator().manual_seed(cout + D)
    x = bf16_round(torch.randn(2, D, H, W, 64, generator=g)).cuda().to(dtype)
    sk = bf16_round(torch.randn(2, D, H, W, cout, generator=g)).cuda().to(dtype)
    w = bf16_round(torch.randn(cout, 64, 3, 3, 3, generator=g) / np.sqrt(27 * 64))
    bn = (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1, torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5)
    layer = ops.Conv3dLayer.build(w, kind=L.CONV_S1, device="cuda", dtype=dtype, bn=bn, relu=True)
    res = {}
    L.set_tuning("conv_small_tiles", 0)
    L.set_tuning("conv_wide", 0)               # (both arms on the brick kernel)
    try:
        for tall in (0, 2):
            L.set_tuning("conv_tall64", tall)
            res[tall] = (ops.conv3d(x, layer, skip=sk), ops.conv3d(x, layer, out_dtype=torch.float32))
    finally:
        L.set_tuning("conv_tall64", 1)
        L.set_tuning("conv_small_tiles", 1)
        L.set_tuning("conv_wide", 1)
    assert torch.equal(res[0][0], res[2][0]) and torch.equal(res[0][1], res[2][1])
    scale = bn[0] / torch.sqrt(bn[3] + 1e-5)
    ref = F.relu(F.conv3d(x.float().cpu().permute(0, 4, 1, 2, 3), w, padding=1) * scale.view(1, -1, 1, 1, 1) + (bn[1] - bn[2] * scale).view(1, -1, 1, 1, 1))
    check_close(f"tall tiles 64->{cout} {dtype}", res[2][1].permute(0, 4, 1, 2, 3).cpu(), ref, max_abs=3e-3, rel_l2=2e-4)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape,slots", [((16, 16, 28), 0), ((13, 11, 21), 0), ((2, 3, 5), 0), ((40, 9, 37), 0), ((37, 20, 33), 4096),
                                         ((9, 64, 70), 0), ((1, 8, 14), 0)])
def test_fused_residual_block_equals_the_two_launches(env, shape, slots, dtype):
    """pscv_conv3d_block8 (BasicBlock of the Vis 3-D U-Net, nn_utils.py:27-37, in one depth sweep with the intermediate volume
    in LDS) against conv3d(layer 1) + conv3d(layer 2, skip = x): same bits.  Ragged sizes (tiles of 8 x 14 cut on both axes, odd
    depth), many depth chunks (block8_slots: halo planes recomputed at every seam), channel slices in and out."""
    L, ops = env
    D, H, W = shape
    g = torch.Generator().manual_seed(D * 100 + W)
    B = 2
    xw = torch.full((B, D, H, W, 16), float("nan"), dtype=dtype, device="cuda")
    x = bf16_round(torch.randn(B, D, H, W, 8, generator=g)).cuda().to(dtype)
    xw[..., 8:] = x
    mk = lambda relu, relu_post: ops.Conv3dLayer.build(
        bf16_round(torch.randn(8, 8, 3, 3, 3, generator=g) / np.sqrt(27 * 8)), kind=L.CONV_S1P8, device="cuda", dtype=dtype,
        bn=(torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.1, torch.randn(8, generator=g) * 0.1, torch.rand(8, generator=g) + 0.5),
        relu=relu, relu_post=relu_post)
    l1, l2 = mk(True, False), mk(False, True)
    want = ops.conv3d(ops.conv3d(x, l1), l2, skip=x)
    L.set_tuning("block8_slots", slots)
    try:
        got = ops.conv3d_block8(x, l1, l2, residual=True)
        out = torch.full((B, D, H, W, 12), -5.0, dtype=dtype, device="cuda")
        ops.conv3d_block8(xw, l1, l2, residual=True, in_coff=8, out=out, out_coff=4)
        plain = ops.conv3d_block8(x, l1, l2, residual=False)
    finally:
        L.set_tuning("block8_slots", 0)
    assert got is not None and torch.equal(got, want)
    assert torch.equal(out[..., 4:], want) and bool((out[..., :4] == -5.0).all())
    assert torch.equal(plain, ops.conv3d(ops.conv3d(x, l1), l2))
    # layers outside the depth-sweep family are declined
    l16 = ops.Conv3dLayer.build(torch.zeros(16, 8, 3, 3, 3), kind=L.CONV_S2, device="cuda", dtype=dtype)
    assert ops.conv3d_block8(x, l1, l16) is None


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("transposed", [False, True])
@pytest.mark.parametrize("cin,cout,shape", [(64, 64, (6, 10, 21)), (64, 32, (9, 17, 40)), (32, 64, (4, 8, 16)), (32, 32, (5, 7, 33)),
                                            (64, 64, (4, 64, 80))])
def test_wide_kernel_equals_the_brick_kernel_and_aten(env, cin, cout, shape, transposed, dtype):
    """Round 5: `conv3d_wide_kernel` (32 | 64 -> 32 | 64 stride-1 layers of CVP's regulariser on large volumes: 8-wave workgroups, the
    weights through an LDS double buffer shared by the waves) against the brick kernel (`conv_wide` = 0): the same k-step order per
    accumulator and the same epilogue chain -> IDENTICAL stored bits, 16-bit and fp32 outputs, with BatchNorm + ReLU + skip, ragged
    sizes, batch of two, stride-1 transposed layers (CVP's conv5: the flipped kernel through the packing).  `conv_wide` = 2 runs the
    wide kernel at any size (by default volumes with fewer than 512 tiles keep the brick kernel).  And against ATen."""
    L, ops = env
    D, H, W = shape
    g = torch.Generator().manual_seed(cin + cout + D + int(transposed))
    x = bf16_round(torch.randn(2, D, H, W, cin, generator=g)).cuda().to(dtype)
    sk = bf16_round(torch.randn(2, D, H, W, cout, generator=g)).cuda().to(dtype)
    w = bf16_round(torch.randn(*((cin, cout) if transposed else (cout, cin)), 3, 3, 3, generator=g) / np.sqrt(27 * cin))
    bn = (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1, torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5)
    layer = ops.Conv3dLayer.build(w, kind=L.CONV_S1, transposed=transposed, device="cuda", dtype=dtype, bn=bn, relu=True)
    res = {}
    L.set_tuning("conv_small_tiles", 0)       # the brick kernel's row-split variant (the k-split small tiles sum in another order)
    try:
        for wide in (0, 2):                   # brick | the wide kernel at any size
            L.set_tuning("conv_wide", wide)
            res[wide] = (ops.conv3d(x, layer, skip=sk), ops.conv3d(x, layer, out_dtype=torch.float32), ops.conv3d(x, layer))
    finally:
        L.set_tuning("conv_wide", 1)
        L.set_tuning("conv_small_tiles", 1)
    for arm in (2,):
        for a_, b_ in zip(res[0], res[arm]):
            assert torch.isfinite(b_.float()).all()
            ne = int((a_ != b_).sum())
            assert ne == 0, f"conv_wide={arm}: {ne} of {a_.numel()} values differ (max {float((a_.float() - b_.float()).abs().max()):.3e})"
    scale = bn[0] / torch.sqrt(bn[3] + 1e-5)
    xc = x.float().cpu().permute(0, 4, 1, 2, 3)
    conv = F.conv_transpose3d(xc, w, stride=1, padding=1) if transposed else F.conv3d(xc, w, padding=1)
    ref = F.relu(conv * scale.view(1, -1, 1, 1, 1) + (bn[1] - bn[2] * scale).view(1, -1, 1, 1, 1))
    check_close(f"wide kernel {cin}->{cout} transposed={transposed} {dtype}", res[2][1].permute(0, 4, 1, 2, 3).cpu(), ref, max_abs=3e-3, rel_l2=2e-4)
