"""GPU: pscv_conv2d (MFMA 2-D convolution of the feature extractor) against ATen conv2d on the same 16-bit-rounded
operands, every layer shape of MVSNet's FeatureNet, ragged sizes, both storage formats, fused BN/ReLU epilogue."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from tests._util import check_close


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from wild_deep_mvs_amd import _lib as L, ops
    L.lib()
    return L, ops


LAYERS = [(3, 8, 3, 1), (8, 8, 3, 1), (8, 16, 5, 2), (16, 16, 3, 1), (16, 32, 5, 2), (32, 32, 3, 1), (16, 32, 3, 1), (32, 16, 3, 1)]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("ci,co,ks,stride", LAYERS)
@pytest.mark.parametrize("B,H,W", [(2, 40, 72), (1, 37, 53)])
def test_conv2d_matches_aten(env, ci, co, ks, stride, B, H, W, dtype):
    L, ops = env
    g = torch.Generator().manual_seed(ci * 100 + co + ks)
    x = torch.randn(B, ci, H, W, generator=g)
    w = torch.randn(co, ci, ks, ks, generator=g) / (ci * ks * ks) ** 0.5
    gamma, beta = torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.1
    mean, var = torch.randn(co, generator=g) * 0.1, torch.rand(co, generator=g) + 0.5
    xr, wr = x.to(dtype).float(), w.to(dtype).float()
    ref = F.conv2d(xr, wr, stride=stride, padding=ks // 2)
    ref = F.relu(F.batch_norm(ref, mean, var, gamma, beta, False, 0.0, 1e-5))
    layer = ops.Conv2dLayer.build(w, stride=stride, device="cuda", bn=(gamma, beta, mean, var), relu=True, dtype=dtype)
    cpad = layer.c_in
    xcl = torch.zeros(B, H, W, cpad, dtype=dtype)
    xcl[..., :ci] = x.permute(0, 2, 3, 1).to(dtype)
    got = ops.conv2d(xcl.cuda(), layer, out_dtype=torch.float32)
    assert tuple(got.shape) == (B, ref.shape[2], ref.shape[3], co)
    check_close(f"conv2d {ci}->{co} k{ks}s{stride} {dtype}", got.permute(0, 3, 1, 2).cpu(), ref,
                max_abs=3e-5 * float(ref.abs().max()) + 1e-6, rel_l2=1e-5)
    # 16-bit output = the fp32 result rounded once
    got16 = ops.conv2d(xcl.cuda(), layer)
    assert got16.dtype == dtype
    ulp = 2 ** -8 if dtype == torch.bfloat16 else 2 ** -11
    check_close("16-bit store", got16.float().cpu(), got.cpu(), max_abs=ulp * float(got.abs().max()) + 1e-6, rel_l2=ulp)


@pytest.mark.parametrize("ci,co,wlds", [(3, 64, 1), (64, 64, 0), (64, 64, 2), (64, 32, 0), (64, 32, 2), (32, 16, 0), (32, 16, 2), (32, 32, 0),
                                        (32, 32, 2), (16, 16, 1)])
def test_conv2d_leaky_relu_layers_of_the_cvp_pyramid(env, ci, co, wlds):
    """conv 3x3 + bias + LeakyReLU(0.1) (models/CVP_MVSNet/models/modules.py:24-28), incl. the 64-channel layers on both of
    their kernels: weight fragments streamed per wave through the prefetch ring (conv2d_wlds = 0) and the persistent kernel
    with the layer's weights resident in LDS (2 = at any size; ragged 45 x 70 maps: partial tiles, 2 x 12 tiles over the CUs)."""
    L, ops = env
    L.set_tuning("conv2d_wlds", wlds)
    g = torch.Generator().manual_seed(ci + co)
    B, H, W = 2, 45, 70
    x = torch.randn(B, ci, H, W, generator=g)
    w = torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5
    bias = torch.randn(co, generator=g) * 0.2
    dtype = torch.float16
    ref = F.leaky_relu(F.conv2d(x.to(dtype).float(), w.to(dtype).float(), bias, padding=1), 0.1)
    layer = ops.Conv2dLayer.build(w, stride=1, device="cuda", conv_bias=bias, leaky=0.1, dtype=dtype)
    xcl = torch.zeros(B, H, W, layer.c_in, dtype=dtype)
    xcl[..., :ci] = x.permute(0, 2, 3, 1).to(dtype)
    try:
        got = ops.conv2d(xcl.cuda(), layer, out_dtype=torch.float32)
        got16 = ops.conv2d(xcl.cuda(), layer)
    finally:
        L.set_tuning("conv2d_wlds", 1)
    check_close(f"leaky conv2d {ci}->{co}", got.permute(0, 3, 1, 2).cpu(), ref, max_abs=3e-5 * float(ref.abs().max()) + 1e-6, rel_l2=1e-5)
    assert torch.equal(got16, got.to(dtype))
    assert float((ref < 0).float().mean()) > 0.2     # the negative branch is exercised


def test_conv2d_plain_conv_with_bias_at_feature_size(env):
    """The extractor's last layer (plain Conv2d 32->32 with bias, no BN / ReLU) at the headline feature size, and the
    linearity of the layer (a size-independent property): conv(a x + b y) = a conv(x) + b conv(y) - (a + b - 1) bias."""
    L, ops = env
    g = torch.Generator().manual_seed(5)
    B, H, W, c = 5, 128, 160, 32
    w = torch.randn(c, c, 3, 3, generator=g) / (c * 9) ** 0.5
    bias = torch.randn(c, generator=g) * 0.1
    layer = ops.Conv2dLayer.build(w, stride=1, device="cuda", conv_bias=bias, relu=False, dtype=torch.float16)
    x = torch.randn(B, H, W, c, generator=g).to(torch.float16).cuda()
    y = torch.randn(B, H, W, c, generator=g).to(torch.float16).cuda()
    fx, fy = ops.conv2d(x, layer, out_dtype=torch.float32), ops.conv2d(y, layer, out_dtype=torch.float32)
    z = (0.5 * x.float() + 0.25 * y.float()).to(torch.float16)          # exactly representable combination
    fz = ops.conv2d(z, layer, out_dtype=torch.float32)
    want = 0.5 * fx + 0.25 * fy + 0.25 * bias.cuda().view(1, 1, 1, c)
    check_close("linearity", fz.cpu(), want.cpu(), max_abs=2e-3 * float(want.abs().max()), rel_l2=1e-3)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.to(torch.float16).float().cuda(), bias.cuda(), padding=1).permute(0, 2, 3, 1)
    check_close("vs ATen on the GPU", fx.cpu(), ref.cpu(), max_abs=1e-4 * float(ref.abs().max()), rel_l2=2e-5)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("ci,co", [(32, 32), (64, 64), (32, 16)])
def test_weights_in_lds_kernel_with_residual_equals_streaming_kernel(env, ci, co, dtype):
    """The persistent weights-in-LDS kernel (conv2d_wlds = 2) on a residual block's second conv (BN + skip + ReLU, the Vis
    extractor's blocks, nn_utils.py:60-100) against the streaming kernel (0): same bits; skip and output as channel slices of
    wider tensors; a ragged size with partial tiles on both axes."""
    L, ops = env
    g = torch.Generator().manual_seed(ci * 7 + co)
    B, H, W = 2, 45, 70
    x = torch.randn(B, H, W, ci, generator=g).to(dtype).cuda()
    w = torch.randn(co, ci, 3, 3, generator=g) / (ci * 9) ** 0.5
    bn = (torch.rand(co, generator=g) + 0.5, torch.randn(co, generator=g) * 0.1, torch.randn(co, generator=g) * 0.1, torch.rand(co, generator=g) + 0.5)
    layer = ops.Conv2dLayer.build(w, stride=1, device="cuda", bn=bn, relu=True, dtype=dtype)
    skip_wide = torch.randn(B, H, W, co + 8, generator=g).to(dtype).cuda()
    res = {}
    for wl in (0, 2):
        L.set_tuning("conv2d_wlds", wl)
        try:
            out = torch.full((B, H, W, co + 16), -3.0, dtype=dtype, device="cuda")
            ops.conv2d(x, layer, skip=skip_wide, skip_coff=8, out=out, out_coff=8)
            res[wl] = (out, ops.conv2d(x, layer, skip=skip_wide[..., 8:].contiguous()), ops.conv2d(x, layer))
        finally:
            L.set_tuning("conv2d_wlds", 1)
    for a, b in zip(res[0], res[2]):
        assert torch.equal(a, b)
    out = res[2][0]
    assert bool((out[..., :8] == -3.0).all()) and bool((out[..., 8 + co:] == -3.0).all())
    assert torch.equal(out[..., 8:8 + co], res[2][1])
    scale = bn[0] / torch.sqrt(bn[3] + 1e-5)
    ref = F.relu(F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w.to(dtype).float(), padding=1) * scale.view(1, -1, 1, 1)
                 + (bn[1] - bn[2] * scale).view(1, -1, 1, 1) + skip_wide[..., 8:].float().cpu().permute(0, 3, 1, 2))
    ulp = 2 ** -8 if dtype == torch.bfloat16 else 2 ** -11
    check_close(f"residual conv2d {ci}->{co} {dtype}", res[2][1].float().cpu().permute(0, 3, 1, 2), ref, max_abs=2 * ulp * float(ref.abs().max()) + 1e-5, rel_l2=2 * ulp)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 3, 64, 80), (1, 3, 37, 50), (3, 3, 2, 2), (1, 3, 75, 100)])
def test_image_prep_equals_interpolate_and_conversion(env, shape, dtype):
    """pscv_image_prep: the extractor input layout (3 channels + 5 zero channels, 16-bit) and the half-scale bilinear level of
    CVP's image pyramid (net.py:44) have the bits of F.interpolate + the torch conversion; ops.image_pyramid_cl8 chains the
    levels (odd widths take F.interpolate for that step)."""
    L, ops = env
    B, C, H, W = shape
    g = torch.Generator().manual_seed(H * W)
    x = (torch.randn(shape, generator=g) * 1.5).cuda()
    want0 = torch.zeros(B, H, W, 8, dtype=dtype, device="cuda")
    want0[..., :C] = x.permute(0, 2, 3, 1)
    assert torch.equal(ops.image_to_channels_last8(x, dtype), want0)
    levels = ops.image_pyramid_cl8(x, 4 if min(H, W) >= 16 else 2, dtype)
    img = x
    for l, got in enumerate(levels):
        if l:
            img = F.interpolate(img, scale_factor=0.5, mode="bilinear", align_corners=None)
        want = torch.zeros(img.shape[0], img.shape[2], img.shape[3], 8, dtype=dtype, device="cuda")
        want[..., :C] = img.permute(0, 2, 3, 1)
        assert got.shape == want.shape and torch.equal(got, want), f"level {l}"
    # slices of one batch tensor are batched without a copy
    imgs = torch.randn(1, 4, 3, 8, 10, device="cuda")
    v = ops.batch_views([imgs[:, i] for i in range(4)])
    assert v.data_ptr() == imgs.data_ptr() and torch.equal(v, torch.cat([imgs[:, i] for i in range(4)], 0))
    two = torch.randn(2, 4, 3, 8, 10, device="cuda")
    assert torch.equal(ops.batch_views([two[:, i] for i in range(4)]), torch.cat([two[:, i] for i in range(4)], 0))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_device_conv2d_weight_packing_equals_host_packing(dtype):
    """pscv_pack_conv2d_weights_device (what Conv2dLayer.build uses for device weights: no host round trip per layer of a training
    step) writes the same bits as the host loop, for every (c_in, padded c_in, c_out, k) family the extractors use."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from wild_deep_mvs_amd import ops
    import numpy as np
    g = torch.Generator().manual_seed(2)
    for (co, ci, k) in [(8, 3, 3), (16, 8, 5), (32, 32, 3), (64, 64, 3), (32, 64, 1), (20, 16, 2), (128, 64, 3)]:
        w = torch.randn(co, ci, k, k, generator=g)
        host = torch.from_numpy(ops.pack_conv2d_weights(w, (ci + 7) // 8 * 8, dtype).view(np.int16))
        dev = ops.Conv2dLayer.build(w.cuda(), stride=1, dtype=dtype).packed
        assert dev.is_cuda and torch.equal(dev.cpu(), host), (co, ci, k)
