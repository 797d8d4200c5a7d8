"""HIP fused warp + cost kernel vs the CPU oracle and the reference goldens (through the C ABI)."""
import numpy as np
import pytest
import torch

from _util import bf16_round, check_close, load_golden, t

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from wild_deep_mvs_amd import _lib as L, ops
    from oracle import mvsnet as O
    return L, ops, O


def _cl(x, dtype):
    from wild_deep_mvs_amd import ops
    return ops.to_channels_last(x.cuda(), dtype)


@pytest.mark.parametrize("fname", ["mvsnet_tiny.npz", "mvsnet_behind.npz", "mvsnet_dtu_tiny.npz"])
def test_warp_only_fp32_matches_reference_golden(env, fname):
    L, ops, O = env
    g = load_golden(fname)
    feats, proj, dv = t(g["features"]), t(g["proj"]), t(g["depth_values"])[:, 0].contiguous()
    V = feats.shape[0]
    planes = g["warped_planes"].tolist()
    cams = ops.proj_cams([proj[:, i].cuda() for i in range(1, V)], proj[:, 0].cuda())
    vol = ops.warp_cost(None, [_cl(feats[i], torch.float32) for i in range(1, V)], cams, dv.cuda(),
                        cost=L.COST_WARP_ONLY, out_dtype=torch.float32)          # [n,B,D,h,w,C]
    for i in range(V - 1):
        got = vol[i].permute(0, 4, 1, 2, 3)[:, :, planes].cpu()
        check_close(f"{fname} warp view {i + 1} vs reference", got, t(g["warped"][i]), max_abs=3e-4)
        ref_full = O.homo_warping(feats[i + 1], proj[:, i + 1], proj[:, 0], dv, feats[0].shape[-2:])
        check_close(f"{fname} warp view {i + 1} vs oracle (all planes)", vol[i].permute(0, 4, 1, 2, 3).cpu(), ref_full,
                    max_abs=3e-4)


def test_homo_warping_function_api(env):
    """models.MVSNet.module.homo_warping(src_fea, src_proj, ref_proj, depth_values, ref_shape) drop-in,
    per-batch and per-pixel depth planes."""
    L, ops, O = env
    from wild_deep_mvs_amd.models.MVSNet.module import homo_warping
    g = load_golden("mvsnet_tiny.npz")
    feats, proj, dv = t(g["features"]), t(g["proj"]), t(g["depth_values"])[:, 0].contiguous()
    planes = g["warped_planes"].tolist()
    w = homo_warping(feats[1].cuda(), proj[:, 1].cuda(), proj[:, 0].cuda(), dv.cuda(), feats[0].shape[-2:])
    assert tuple(w.shape) == (1, 32, dv.shape[1]) + tuple(feats[0].shape[-2:])
    check_close("homo_warping per-batch planes", w[:, :, planes].cpu(), t(g["warped"][0]), max_abs=3e-4)
    wpp = homo_warping(feats[1].cuda(), proj[:, 1].cuda(), proj[:, 0].cuda(), t(g["depth_per_pixel"]).cuda(),
                       feats[0].shape[-2:])
    check_close("homo_warping per-pixel planes", wpp[:, :, planes].cpu(), t(g["warped_per_pixel"]), max_abs=3e-4)
    from wild_deep_mvs_amd.models import utils as mutils          # the name BASELINE.json gives the warp
    assert mutils.homo_warp is homo_warping


@pytest.mark.parametrize("fname,agg", [("mvsnet_tiny.npz", "variance"), ("mvsnet_behind.npz", "variance"),
                                        ("mvsnet_dtu_tiny.npz", "variance"), ("mvsnet_s_tiny.npz", "softmin")])
def test_cost_volume_fp32_matches_reference_golden(env, fname, agg):
    L, ops, O = env
    g = load_golden(fname)
    feats, proj, dv = t(g["features"]), t(g["proj"]), t(g["depth_values"])[:, 0].contiguous()
    V = feats.shape[0]
    cams = ops.proj_cams([proj[:, i].cuda() for i in range(1, V)], proj[:, 0].cuda())
    cost = ops.warp_cost(_cl(feats[0], torch.float32), [_cl(feats[i], torch.float32) for i in range(1, V)], cams,
                         dv.cuda(), cost=L.COST_VARIANCE if agg == "variance" else L.COST_SOFTMIN, temp=1.0,
                         out_dtype=torch.float32)
    ref = t(g["cost_volume"])
    check_close(f"{fname} {agg} cost volume fp32", cost.permute(0, 4, 1, 2, 3).cpu(), ref,
                max_abs=2e-4 * max(1.0, float(ref.abs().max())), rel_l2=2e-4)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("lpv", [0, 4, 2, 1, -1])
def test_cost_volume_16bit_storage(env, lpv, dtype):
    """16-bit feature maps in, 16-bit cost volume out, fp32 accumulation: equals the oracle run on the
    rounded features up to one output rounding."""
    L, ops, O = env
    g = load_golden("mvsnet_tiny.npz")
    feats, proj, dv = t(g["features"]), t(g["proj"]), t(g["depth_values"])[:, 0].contiguous()
    V = feats.shape[0]
    fr = [f.to(dtype).float() for f in feats]
    warped = [O.homo_warping(fr[i], proj[:, i], proj[:, 0], dv, fr[0].shape[-2:]) for i in range(1, V)]
    ref = O.variance_cost(fr[0], warped)
    cams = ops.proj_cams([proj[:, i].cuda() for i in range(1, V)], proj[:, 0].cuda())
    # 0 = default mapping; 4/2/1 = direct kernel with that many lanes per voxel; -1 = LDS-staged tiled kernel
    L.set_tuning("warp_lpv", max(lpv, 0))
    L.set_tuning("warp_tiled", 1 if lpv < 0 else 0)
    try:
        cost = ops.warp_cost(_cl(feats[0], dtype), [_cl(feats[i], dtype) for i in range(1, V)],
                             cams, dv.cuda(), cost=L.COST_VARIANCE, out_dtype=dtype)
    finally:
        L.set_tuning("warp_lpv", 0)
        L.set_tuning("warp_tiled", -1)
    ulp = 2 ** -8 if dtype == torch.bfloat16 else 2 ** -11
    s = check_close(f"variance cost {dtype} storage lpv={lpv}", cost.float().permute(0, 4, 1, 2, 3).cpu(), ref, rel_l2=ulp)
    assert s["max_abs"] <= ulp * s["ref_max"] + 3e-4


def test_device_camera_blocks_match_host_math(env):
    """pscv_proj_cams (one launch, fp64 inside) == the host fp64 closed form, any reference frame."""
    L, ops, O = env
    g = load_golden("mvsnet_behind.npz")
    proj = t(g["proj"])
    V = proj.shape[1]
    for ref in (0, 2):
        src = [i for i in range(V) if i != ref]
        host = ops.proj_cams([proj[:, i] for i in src], proj[:, ref])
        dev = ops.proj_cams_device(proj.cuda().contiguous(), ref).cpu()
        check_close(f"device cams ref={ref}", dev, host, max_abs=1e-6 * float(host.abs().max()))


def test_cvp_variance_rounding_and_16_channels(env):
    """16-channel features (CVP) and the CVP rounding order sum f^2/N - (sum f/N)^2."""
    L, ops, O = env
    g = load_golden("mvsnet_tiny.npz")
    feats, proj, dv = t(g["features"])[:, :, :16].contiguous(), t(g["proj"]), t(g["depth_values"])[:, 0].contiguous()
    V = feats.shape[0]
    warped = [O.homo_warping(feats[i], proj[:, i], proj[:, 0], dv, feats[0].shape[-2:]) for i in range(1, V)]
    N = V
    s = feats[0].unsqueeze(2) + sum(warped)
    sq = feats[0].unsqueeze(2) ** 2 + sum(w ** 2 for w in warped)
    ref = sq / N - (s / N) ** 2
    cams = ops.proj_cams([proj[:, i].cuda() for i in range(1, V)], proj[:, 0].cuda())
    cost = ops.warp_cost(_cl(feats[0], torch.float32), [_cl(feats[i], torch.float32) for i in range(1, V)], cams,
                         dv.cuda(), cost=L.COST_VARIANCE_CVP, out_dtype=torch.float32)
    check_close("cvp variance 16ch", cost.permute(0, 4, 1, 2, 3).cpu(), ref, max_abs=3e-4, rel_l2=2e-4)


def test_identity_sweep_has_zero_variance_at_full_size(env):
    """Size-independent property at the headline size (5 views, 128x160 features, D=192): when every
    source view IS the reference view (same camera, same features) the warp is the identity and the
    variance cost is exactly sum f^2/N - (N f)^2/N^2 = 0 up to fp32 rounding."""
    L, ops, O = env
    from wild_deep_mvs_amd import synthetic
    B, V, C, h, w, D = 1, 5, 32, 128, 160, 192
    f = synthetic.make_features(B, 1, C, h, w, seed=3)[0].cuda()
    fcl = ops.to_channels_last(f, torch.bfloat16)
    cam = synthetic.make_cameras(B, 1, 4 * h, 4 * w)
    from oracle.mvsnet import mvsnet_cameras
    proj, dvals = mvsnet_cameras(cam["K"], cam["R"], cam["t"], cam["depth_min"], cam["depth_max"], D)
    P = proj[:, 0].cuda()
    cams = ops.proj_cams([P] * (V - 1), P)
    cost = ops.warp_cost(fcl, [fcl] * (V - 1), cams, dvals[:, 0].contiguous().cuda(), cost=L.COST_VARIANCE,
                         out_dtype=torch.float32)
    assert tuple(cost.shape) == (B, D, h, w, C)
    scale = float(fcl.float().pow(2).max())
    assert float(cost.abs().max()) <= 2e-5 * scale, float(cost.abs().max())
    # and the plain warp reproduces the feature map on every plane
    vol = ops.warp_cost(None, [fcl], cams[:1], dvals[:, 0].contiguous().cuda(), cost=L.COST_WARP_ONLY,
                        out_dtype=torch.float32)
    assert float((vol[0] - fcl.float().unsqueeze(1)).abs().max()) <= 1e-4 * float(fcl.float().abs().max())


@pytest.mark.parametrize("cost_name", ["variance", "softmin", "variance_cvp"])
@pytest.mark.parametrize("baseline_scale,shape,V,D", [(1.0, (64, 80), 5, 24), (1.0, (37, 53), 5, 24), (12.0, (64, 80), 5, 24),
                                                      (1.0, (40, 56), 2, 7), (3.0, (48, 48), 4, 50)])
def test_tiled_kernel_equals_direct_kernel(env, baseline_scale, shape, V, D, cost_name):
    """The LDS-staged kernel (fp32 patches converted while they are staged, full-rate fp32 blend) and the direct-gather
    kernel run the same fp32 operation chain on the same taps, so their cost volumes agree to the last stored bit (one
    ulp allowed: the compiler may contract the final variance / softmin expression differently): small epipolar spans
    (everything staged), tile sizes that do not divide the image, boxes clipped at the image border and boxes entirely
    outside it (zero padding), a 12x wider baseline where boxes overflow the LDS budget or have corners behind the camera
    and views fall back to direct taps inside the staged kernel, one source view, odd and > 24 plane counts."""
    L, ops, O = env
    from wild_deep_mvs_amd import synthetic
    from oracle.mvsnet import mvsnet_cameras
    h, w = shape
    B, C = 2, 32
    feats = synthetic.make_features(B, V, C, h, w, seed=11)
    cam = synthetic.make_cameras(B, V, 4 * h, 4 * w)
    cam["t"] = cam["t"] * baseline_scale
    proj, dvals = mvsnet_cameras(cam["K"], cam["R"], cam["t"], cam["depth_min"], cam["depth_max"], D)
    cams = ops.proj_cams_device(proj.cuda().contiguous(), 0)
    fcl = [ops.to_channels_last(feats[i].cuda(), torch.float16) for i in range(V)]
    dv = dvals[:, 0].contiguous().cuda()
    code = {"variance": L.COST_VARIANCE, "softmin": L.COST_SOFTMIN, "variance_cvp": L.COST_VARIANCE_CVP}[cost_name]
    outs = []
    for tiled in (1, 0):
        L.set_tuning("warp_tiled", tiled)
        try:
            outs.append(ops.warp_cost(fcl[0], fcl[1:], cams, dv, cost=code, temp=0.7, out_dtype=torch.float16).float().cpu())
        finally:
            L.set_tuning("warp_tiled", -1)
    s = check_close(f"tiled vs direct {cost_name} baseline x{baseline_scale} {shape}", outs[0], outs[1],
                    max_abs=2 ** -10 * float(outs[1].abs().max()), rel_l2=2e-5)
    assert float(outs[1].abs().max()) > 0


@pytest.mark.parametrize("D", [64, 44, 192])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_adaptive_split_levels_store_the_same_bits_on_the_wide_baseline_rig(env, dtype, D):
    """The LDS-staged kernel on the DTU-like rig (0.14-0.32 texels per plane: 32-plane boxes overflow the arena) with the adaptive
    split at its three settings -- `warp_tile` 0: halves, and quarters where a half still has a view on global taps (round 6);
    3: halves only (round 5); 1: no split -- against the direct-gather kernel: the same stored bits in every case (which planes a
    staging phase covers never changes the fp32 chain of a voxel), and every split level lowers the share of (range, view) pairs that
    take global taps.  D = 44: a last chunk of 12 planes (halves of 6, quarters of 4 + 2); D = 192: the headline's six chunks."""
    import ctypes
    L, ops, O = env
    from wild_deep_mvs_amd import synthetic
    from wild_deep_mvs_amd.models.MVSNet.model import build_proj_matrices
    V, h, w, C = 5, 128, 160, 32
    feats = synthetic.make_features(1, V, C, h, w, seed=4)
    fcl = [ops.to_channels_last(feats[i].cuda(), dtype) for i in range(V)]
    cam = synthetic.make_cameras(1, V, 4 * h, 4 * w, rig="dtu")
    Ks = cam["K"].clone()
    Ks[:, :, :2] /= 4
    proj = build_proj_matrices(Ks, cam["R"], cam["t"]).cuda()
    dv = torch.linspace(float(cam["depth_min"][0, 0]), float(cam["depth_max"][0, 0]), D).view(1, D).cuda()
    cams = ops.proj_cams_device(proj.float().contiguous(), 0)
    run = lambda: ops.warp_cost(fcl[0], fcl[1:], cams, dv, cost=L.COST_VARIANCE, out_dtype=dtype)
    fn = L.lib().pscv_debug_wl_mode_hist
    fn.argtypes, fn.restype = [ctypes.c_void_p], None
    try:
        L.set_tuning("warp_tiled", 0)
        want = run().clone()
        L.set_tuning("warp_tiled", 1)
        direct_share = {}
        for tile in (0, 3, 1):
            L.set_tuning("warp_tile", tile)
            hist = torch.zeros(16, dtype=torch.int32, device="cuda")
            fn(hist.data_ptr())
            try:
                got = run()
                torch.cuda.synchronize()
            finally:
                fn(None)
            assert torch.equal(got, want), f"warp_tile = {tile}: stored bits differ from the direct-gather kernel"
            hm = hist.view(4, 4).cpu()                       # [view][DIRECT, GEN, FAST, ZERO]
            direct_share[tile] = float(hm[:, 0].sum()) / float(hm.sum())
    finally:
        L.set_tuning("warp_tiled", -1)
        L.set_tuning("warp_tile", 0)
    print(f"[parity] DTU-like rig D={D} {dtype}: share of (plane range, view) pairs on global taps: no split {direct_share[1]:.3f}, "
          f"halves {direct_share[3]:.3f}, halves + quarters {direct_share[0]:.3f}", flush=True)
    assert direct_share[0] < direct_share[3] < direct_share[1]
    if D == 192:           # the headline's plane spacing (0.14-0.32 texels per plane; 48-plane chunks, 12-plane quarters): measured 0.63 / 0.30 / 0.08
        assert direct_share[0] <= 0.10 and direct_share[3] >= 0.15


@pytest.mark.parametrize("dtype,out_dtype", [(torch.float16, torch.float16), (torch.bfloat16, torch.bfloat16), (torch.float16, torch.float32)])
@pytest.mark.parametrize("cost_name", ["variance", "variance_cvp"])
@pytest.mark.parametrize("baseline_scale,shape,V,D", [(1.0, (64, 80), 5, 24), (1.0, (37, 53), 5, 24), (12.0, (64, 80), 5, 24),
                                                      (1.0, (40, 56), 2, 7), (3.0, (48, 48), 4, 50), (1.0, (128, 160), 5, 64)])
def test_lane_owner_kernel_equals_direct_kernel(env, baseline_scale, shape, V, D, cost_name, dtype, out_dtype):
    """The lane-owns-voxel kernel (warp_cost_lv.hip, `warp_tiled` = 4: channel-chunk planar boxes with their zero padding staged,
    straight-line sweep per count of staged views, stores transposed across the rows of a wave) stores the SAME BITS as the
    direct-gather kernel: fast path and -- `warp_tile` = 7 -- every block forced onto its general path (per-chunk positions, per-lane
    global taps for views that are not staged).  Cases as for the quad-owner kernel: everything staged, image sizes the 8 x 4 tile
    does not divide, boxes clipped at the border / outside the image, a 12x wider baseline (boxes that do not fit, corners behind the
    camera), one source view, odd plane counts (the unpaired last plane of a trip) and more planes than one chunk; fp32 output
    takes the un-transposed stores."""
    L, ops, O = env
    from wild_deep_mvs_amd import synthetic
    from oracle.mvsnet import mvsnet_cameras
    h, w = shape
    B, C = 2, 32
    feats = synthetic.make_features(B, V, C, h, w, seed=13)
    cam = synthetic.make_cameras(B, V, 4 * h, 4 * w)
    cam["t"] = cam["t"] * baseline_scale
    proj, dvals = mvsnet_cameras(cam["K"], cam["R"], cam["t"], cam["depth_min"], cam["depth_max"], D)
    cams = ops.proj_cams_device(proj.cuda().contiguous(), 0)
    fcl = [ops.to_channels_last(feats[i].cuda(), dtype) for i in range(V)]
    dv = dvals[:, 0].contiguous().cuda()
    code = {"variance": L.COST_VARIANCE, "variance_cvp": L.COST_VARIANCE_CVP}[cost_name]
    outs = {}
    # ("warp_tile" = 2: the adaptive split of a chunk whose boxes do not fit, round 5; off by default in this kernel)
    for name, tiled, tile in (("lane-owner", 4, 0), ("lane-owner, general path", 4, 7), ("lane-owner, split", 4, 2), ("direct", 0, 0)):
        L.set_tuning("warp_tiled", tiled)
        L.set_tuning("warp_tile", tile)
        try:
            outs[name] = ops.warp_cost(fcl[0], fcl[1:], cams, dv, cost=code, out_dtype=out_dtype)
        finally:
            L.set_tuning("warp_tiled", -1)
            L.set_tuning("warp_tile", 0)
    torch.cuda.synchronize()
    want = outs["direct"]
    assert float(want.float().abs().max()) > 0
    for name in ("lane-owner", "lane-owner, general path", "lane-owner, split"):
        ne = int((outs[name] != want).sum())
        assert ne == 0, (f"{name} vs direct, {cost_name} baseline x{baseline_scale} {shape}: {ne} of {want.numel()} values differ, "
                         f"max abs {float((outs[name].float() - want.float()).abs().max()):.3e}")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("cost_name", ["variance", "variance_cvp", "softmin", "warp_only", "groupcorr", "warp_only_homog"])
@pytest.mark.parametrize("baseline_scale,shape,D,per_pixel", [(1.0, (64, 80), 24, False), (1.0, (37, 53), 23, True),
                                                              (12.0, (40, 48), 7, False)])
def test_quad_kernel_equals_generic_kernel(env, baseline_scale, shape, D, per_pixel, cost_name, dtype):
    """The quad-mapped kernel (warp_cost_quad.hip: one texel per lane quad, two depth planes per quad) and the generic
    2-lanes-per-voxel kernel run the same arithmetic on the same taps: every cost mode of both geometries, an odd
    plane count (the unpaired tail plane), image sizes that do not divide the 64-pixel blocks, per-pixel planes (CVP /
    Vis refinement stages), and a 12x wider baseline where most samples leave the image (border path, zero padding,
    points behind the camera).  Agreement to one stored ulp (the compiler contracts the final variance / softmin
    expression differently in the two kernels)."""
    L, ops, O = env
    from wild_deep_mvs_amd import synthetic
    from oracle.mvsnet import mvsnet_cameras
    h, w = shape
    B, V, C = 2, 4, 32
    feats = synthetic.make_features(B, V, C, h, w, seed=5)
    cam = synthetic.make_cameras(B, V, 4 * h, 4 * w)
    cam["t"] = cam["t"] * baseline_scale
    proj, dvals = mvsnet_cameras(cam["K"], cam["R"], cam["t"], cam["depth_min"], cam["depth_max"], D)
    fcl = [ops.to_channels_last(feats[i].cuda(), dtype) for i in range(V)]
    dv = dvals[:, 0].contiguous()
    if per_pixel:
        gen = torch.Generator().manual_seed(1)
        dv = (dv.view(B, D, 1, 1) * (1.0 + 0.05 * torch.rand(B, 1, h, w, generator=gen))).contiguous()
    homog = cost_name in ("groupcorr", "warp_only_homog")
    if homog:
        import oracle.vismvsnet as OV
        di = (cam["depth_max"] - cam["depth_min"]) / D
        arr = [OV.fill_cam_array(cam["K"][:, i], cam["R"][:, i], cam["t"][:, i], cam["depth_min"][:, i], di[:, i]) for i in range(V)]
        cams = ops.homog_cams_device(arr[0].cuda(), [a.cuda() for a in arr[1:]], 1.0 / 4)
        geom = L.GEOM_HOMOG
    else:
        cams = ops.proj_cams_device(proj.cuda().contiguous(), 0)
        geom = L.GEOM_PROJ
    code = {"variance": L.COST_VARIANCE, "variance_cvp": L.COST_VARIANCE_CVP, "softmin": L.COST_SOFTMIN,
            "warp_only": L.COST_WARP_ONLY, "groupcorr": L.COST_GROUPCORR, "warp_only_homog": L.COST_WARP_ONLY}[cost_name]
    ref_in = None if code == L.COST_WARP_ONLY else fcl[0]
    outs = []
    for q2 in (1, 0):
        L.set_tuning("warp_q2", q2)
        L.set_tuning("warp_tiled", 0)      # (the LDS-staged kernel is the default where it applies: compare the two direct kernels)
        try:
            outs.append(ops.warp_cost(ref_in, fcl[1:], cams, dv.cuda(), geom=geom, cost=code, temp=0.7, out_dtype=dtype).float().cpu())
        finally:
            L.set_tuning("warp_q2", 1)
            L.set_tuning("warp_tiled", -1)
    ulp = 2 ** -7 if dtype == torch.bfloat16 else 2 ** -10
    assert float(outs[1].abs().max()) > 0
    check_close(f"quad vs generic {cost_name} {dtype} baseline x{baseline_scale} {shape} D={D}", outs[0], outs[1],
                max_abs=ulp * float(outs[1].abs().max()), rel_l2=ulp / 16)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("baseline_scale,shape,V,D,per_pixel", [(1.0, (64, 80), 5, 24, False), (1.0, (37, 53), 4, 23, True), (12.0, (40, 48), 4, 7, False),
                                                              (1.0, (72, 100), 9, 40, False), (1.0, (48, 64), 3, 32, True), (3.0, (48, 48), 7, 16, True)])
def test_lds_staged_groupcorr_equals_quad_kernel(env, baseline_scale, shape, V, D, per_pixel, dtype):
    """The LDS-staged group-correlation kernel (warp_gc_lv.hip, `warp_gc_lds`) against the quad kernel on the same taps:
    HOMOG geometry with per-batch and per-pixel planes (the Vis-MVSNet stages), 2-8 source views (groups of four per launch), image
    sizes the 8 x 4 tile does not divide, odd plane counts, a wide baseline (clipped boxes, boxes outside the image, views that do not
    fit and take global taps) and -- `warp_tile` = 7 -- nothing staged at all.  The warp runs the same fp32 chain in both kernels;
    the group sums may differ in the last fp32 bit, so the stored values agree to one 16-bit ulp of the volume's scale and nearly all
    are equal."""
    L, ops, O = env
    from wild_deep_mvs_amd import synthetic
    import oracle.vismvsnet as OV
    h, w = shape
    B, C = 2, 32
    feats = synthetic.make_features(B, V, C, h, w, seed=21)
    cam = synthetic.make_cameras(B, V, 4 * h, 4 * w)
    cam["t"] = cam["t"] * baseline_scale
    di = (cam["depth_max"] - cam["depth_min"]) / D
    arr = [OV.fill_cam_array(cam["K"][:, i], cam["R"][:, i], cam["t"][:, i], cam["depth_min"][:, i], di[:, i]) for i in range(V)]
    cams = ops.homog_cams_device(arr[0].cuda(), [a.cuda() for a in arr[1:]], 1.0 / 4)
    fcl = [ops.to_channels_last(feats[i].cuda(), dtype) for i in range(V)]
    dv = (cam["depth_min"][:, :1] + di[:, :1] * torch.arange(D, dtype=torch.float32).view(1, D)).contiguous()
    if per_pixel:
        gen = torch.Generator().manual_seed(1)
        dv = (dv.view(B, D, 1, 1) * (1.0 + 0.05 * torch.rand(B, 1, h, w, generator=gen))).contiguous()
    outs = {}
    assert L.get_tuning("warp_gc_lds") == 1          # (default: per-batch planes on the staged kernel, per-pixel planes on the quad kernel)
    for name, gc, tile in (("staged", 2, 0), ("staged kernel, nothing staged", 2, 7), ("quad", 0, 0)):
        L.set_tuning("warp_gc_lds", gc)
        L.set_tuning("warp_tile", tile)
        try:
            outs[name] = ops.warp_cost(fcl[0], fcl[1:], cams, dv.cuda(), geom=L.GEOM_HOMOG, cost=L.COST_GROUPCORR, out_dtype=dtype).float().cpu()
        finally:
            L.set_tuning("warp_gc_lds", 1)
            L.set_tuning("warp_tile", 0)
    want = outs["quad"]
    assert tuple(want.shape) == (V - 1, B, D, h, w, 8) and float(want.abs().max()) > 0
    ulp = 2 ** -7 if dtype == torch.bfloat16 else 2 ** -10
    for name in ("staged", "staged kernel, nothing staged"):
        check_close(f"groupcorr {name} vs quad {dtype} baseline x{baseline_scale} {shape} V={V} D={D} per_pixel={per_pixel}", outs[name], want,
                    max_abs=ulp * float(want.abs().max()), rel_l2=ulp / 16)
        assert float((outs[name] != want).float().mean()) < 0.02, f"{name}: {float((outs[name] != want).float().mean()):.4f} of the values differ"


@pytest.mark.parametrize("seed", list(range(10)))
def test_lds_staged_kernels_on_random_rigs(env, seed):
    """Seeded random draws of what the staged kernels decide on -- image size (tile multiples or not), view and plane counts, baseline
    (0.5x .. 20x: everything staged .. boxes that do not fit, samples behind the camera), camera rig, storage format: the variance
    volume of the quad-owner (default) and lane-owner kernels equals the direct-gather kernel's bit for bit; the group-correlation
    volume of the staged kernel (per-batch and per-pixel planes) agrees with the quad kernel's to one 16-bit ulp."""
    L, ops, O = env
    from wild_deep_mvs_amd import synthetic
    from oracle.mvsnet import mvsnet_cameras
    import oracle.vismvsnet as OV
    rng = np.random.default_rng(1000 + seed)
    B, V = int(rng.integers(1, 3)), int(rng.integers(2, 6))
    h, w = int(rng.integers(21, 90)), int(rng.integers(21, 110))
    D = int(rng.integers(3, 70))
    scale = float(np.exp(rng.uniform(np.log(0.5), np.log(20.0))))
    rig = "dtu" if rng.random() < 0.4 else "probe"
    dtype = torch.bfloat16 if rng.random() < 0.4 else torch.float16
    per_pixel = bool(rng.random() < 0.5)
    feats = synthetic.make_features(B, V, 32, h, w, seed=seed)
    cam = synthetic.make_cameras(B, V, 4 * h, 4 * w, rig=rig)
    cam["t"] = cam["t"] * scale
    proj, dvals = mvsnet_cameras(cam["K"], cam["R"], cam["t"], cam["depth_min"], cam["depth_max"], D)
    fcl = [ops.to_channels_last(feats[i].cuda(), dtype) for i in range(V)]
    tag = f"seed {seed}: B={B} V={V} {h}x{w} D={D} baseline x{scale:.2f} {rig} {dtype} per_pixel={per_pixel}"
    # variance (PROJ geometry, per-batch planes)
    cams = ops.proj_cams_device(proj.cuda().contiguous(), 0)
    dv = dvals[:, 0].contiguous().cuda()
    outs = {}
    for name, tiled in (("quad-owner", 1), ("lane-owner", 4), ("direct", 0)):
        L.set_tuning("warp_tiled", tiled)
        try:
            outs[name] = ops.warp_cost(fcl[0], fcl[1:], cams, dv, cost=L.COST_VARIANCE, out_dtype=dtype)
        finally:
            L.set_tuning("warp_tiled", -1)
    for name in ("quad-owner", "lane-owner"):
        ne = int((outs[name] != outs["direct"]).sum())
        assert ne == 0, f"{tag}: variance, {name} vs direct: {ne} of {outs['direct'].numel()} values differ"
    # group correlation (HOMOG geometry)
    di = (cam["depth_max"] - cam["depth_min"]) / D
    arr = [OV.fill_cam_array(cam["K"][:, i], cam["R"][:, i], cam["t"][:, i], cam["depth_min"][:, i], di[:, i]) for i in range(V)]
    hc = ops.homog_cams_device(arr[0].cuda(), [a.cuda() for a in arr[1:]], 1.0 / 4)
    planes = (cam["depth_min"][:, :1] + di[:, :1] * torch.arange(D, dtype=torch.float32).view(1, D)).contiguous()
    if per_pixel:
        gen = torch.Generator().manual_seed(seed)
        planes = (planes.view(B, D, 1, 1) * (1.0 + 0.03 * torch.rand(B, 1, h, w, generator=gen))).contiguous()
    gouts = {}
    for name, gc in (("staged", 2), ("quad", 0)):
        L.set_tuning("warp_gc_lds", gc)
        try:
            gouts[name] = ops.warp_cost(fcl[0], fcl[1:], hc, planes.cuda(), geom=L.GEOM_HOMOG, cost=L.COST_GROUPCORR, out_dtype=dtype).float().cpu()
        finally:
            L.set_tuning("warp_gc_lds", 1)
    ulp = 2 ** -7 if dtype == torch.bfloat16 else 2 ** -10
    scale_v = max(float(gouts["quad"].abs().max()), 1e-6)
    check_close(f"{tag}: groupcorr staged vs quad", gouts["staged"], gouts["quad"], max_abs=ulp * scale_v, rel_l2=ulp / 16 if scale_v > 1e-3 else None)


def test_staged_kernel_fp16_stores_saturate(env):
    """fp16 cost volumes saturate at +-65504 instead of becoming inf (every kernel of the engine does; the LDS-staged warp
    kernel gets it from the MODE.FP16_OVFL bit instead of a per-element clamp): features of magnitude ~300 give variances up
    to ~9e4.  Same stored bits as the direct kernel (which clamps explicitly)."""
    L, ops, O = env
    from wild_deep_mvs_amd import synthetic
    from oracle.mvsnet import mvsnet_cameras
    B, V, C, h, w, D = 1, 3, 32, 40, 48, 8
    feats = synthetic.make_features(B, V, C, h, w, seed=2) * 600.0
    cam = synthetic.make_cameras(B, V, 4 * h, 4 * w)
    proj, dvals = mvsnet_cameras(cam["K"], cam["R"], cam["t"], cam["depth_min"], cam["depth_max"], D)
    cams = ops.proj_cams_device(proj.cuda().contiguous(), 0)
    fcl = [ops.to_channels_last(feats[i].cuda(), torch.float16) for i in range(V)]
    dv = dvals[:, 0].contiguous().cuda()
    outs = []
    for tiled in (1, 0):
        L.set_tuning("warp_tiled", tiled)
        try:
            outs.append(ops.warp_cost(fcl[0], fcl[1:], cams, dv, cost=L.COST_VARIANCE, out_dtype=torch.float16))
        finally:
            L.set_tuning("warp_tiled", -1)
    assert torch.isfinite(outs[0]).all() and float(outs[0].max()) == 65504.0
    assert (outs[0] == 65504.0).float().mean() > 1e-3            # the case really overflows
    assert torch.equal(outs[0], outs[1])


def test_tuning_knobs_thread_override_and_process_wide(env):
    """pscv_set_tuning_thread steers only the launches of the calling host thread (include/pscv.h): with an unsupported
    lanes-per-voxel value set for this thread its own launch fails loudly, while the same call from another thread runs with the
    process-wide value and gives the default result.  pscv_set_tuning itself is process-wide: a second thread (autograd's backward
    thread, a DataParallel replica) sees it and fails the same way."""
    import threading
    L, ops, O = env
    g = load_golden("mvsnet_tiny.npz")
    feats, proj, dv = t(g["features"]), t(g["proj"]), t(g["depth_values"])[:, 0].contiguous()
    V = feats.shape[0]
    cams = ops.proj_cams([proj[:, i].cuda() for i in range(1, V)], proj[:, 0].cuda())
    fcl = [_cl(feats[i], torch.float16) for i in range(V)]
    run = lambda: ops.warp_cost(fcl[0], fcl[1:], cams, dv.cuda(), cost=L.COST_VARIANCE, out_dtype=torch.float16)
    want = run()
    torch.cuda.synchronize()

    def in_other_thread():
        res = {}

        def other():
            try:
                res["out"] = run()
                torch.cuda.synchronize()
            except Exception as e:     # noqa: BLE001
                res["err"] = e
        th = threading.Thread(target=other)
        th.start(); th.join()
        return res
    L.set_tuning_thread("warp_lpv", 3)
    try:
        with pytest.raises(L.PscvError):
            run()
        res = in_other_thread()
        assert "err" not in res, res.get("err")
        assert torch.equal(res["out"], want)
    finally:
        L.set_tuning_thread("warp_lpv", 0, enable=False)
    assert torch.equal(run(), want)
    L.set_tuning("warp_lpv", 3)
    try:
        res = in_other_thread()
        assert isinstance(res.get("err"), L.PscvError)
    finally:
        L.set_tuning("warp_lpv", 0)
    assert torch.equal(run(), want)


def test_staged_kernel_inf_inputs_stay_inf(env):
    """MODE.FP16_OVFL clamps overflowing FINITE results; a feature map that already holds +inf gives an inf / NaN cost where the
    direct kernel's explicit clamp stores +-65504 (documented exception in csrc/pscv_common.h; unreachable from the engine's own
    16-bit stores, which all saturate).  Everything else stays bit-equal."""
    L, ops, O = env
    from wild_deep_mvs_amd import synthetic
    from oracle.mvsnet import mvsnet_cameras
    B, V, C, h, w, D = 1, 3, 32, 40, 48, 8
    feats = synthetic.make_features(B, V, C, h, w, seed=2)
    feats[0][0, 3, 10, 10] = float("inf")                       # one inf texel in the reference view
    cam = synthetic.make_cameras(B, V, 4 * h, 4 * w)
    proj, dvals = mvsnet_cameras(cam["K"], cam["R"], cam["t"], cam["depth_min"], cam["depth_max"], D)
    cams = ops.proj_cams_device(proj.cuda().contiguous(), 0)
    fcl = [ops.to_channels_last(feats[i].cuda(), torch.float16) for i in range(V)]
    dv = dvals[:, 0].contiguous().cuda()
    outs = []
    for tiled in (1, 0):
        L.set_tuning("warp_tiled", tiled)
        try:
            outs.append(ops.warp_cost(fcl[0], fcl[1:], cams, dv, cost=L.COST_VARIANCE, out_dtype=torch.float16))
        finally:
            L.set_tuning("warp_tiled", -1)
    staged, direct = outs
    bad = ~torch.isfinite(staged)
    assert bad.any() and bad[0, :, 10, 10, 3].all() and int(bad.sum()) == D       # inf - inf = NaN at that voxel column only
    assert torch.equal(staged[~bad], direct[~bad])


@pytest.mark.parametrize("case", ["lds_variance", "lane_owner_variance", "quad_variance", "quad_per_pixel", "generic_16ch", "homog_groupcorr", "homog_groupcorr_lds",
                                  "homog_groupcorr_lds_per_pixel"])
def test_row_slab_launch_equals_the_rows_of_the_whole_image_launch(env, case):
    """`pscv_warp_cost_rows` (ABI 7; the row-sharded Vis-MVSNet stages): a launch on rows [y0, y0 + hs) of the reference grid -- cropped
    reference map and per-pixel planes, the cameras of the WHOLE image, `ref_y0 = y0` -- stores bit for bit the rows of the whole-image
    launch, for every kernel family (LDS-staged, quad, generic; PROJ and HOMOG geometry), at slab origins that are not tile multiples."""
    L, ops, O = env
    from wild_deep_mvs_amd import synthetic
    from oracle.mvsnet import mvsnet_cameras
    B, V, h, w, D = 2, 4, 52, 72, 12
    C = 16 if case == "generic_16ch" else 32
    feats = synthetic.make_features(B, V, C, h, w, seed=5)
    fcl = [ops.to_channels_last(feats[i].cuda(), torch.float16) for i in range(V)]
    kw = {}
    if case.startswith("homog_groupcorr"):
        from oracle import vismvsnet as OV
        sc = synthetic.make_scene(B, V, 4 * h, 4 * w, seed=2)
        di = (sc["depth_max"] - sc["depth_min"]) / 128
        cams_v = [OV.fill_cam_array(sc["K"][:, i], sc["R"][:, i], sc["t"][:, i], sc["depth_min"][:, i], di[:, i]) for i in range(V)]
        cams = ops.homog_cams_device(cams_v[0].cuda(), [c.cuda() for c in cams_v[1:]], 0.25)
        dv = (cams_v[0][:, 1, 3, 0].view(B, 1) + di[:, :1] * 4.0 * torch.arange(D, dtype=torch.float32).view(1, D)).contiguous().cuda()
        if case != "homog_groupcorr_lds":
            dv = (dv.view(B, D, 1, 1) + 0.02 * torch.rand(B, 1, h, w, device="cuda")).contiguous()          # per-pixel starts like stages 2-3
        kw = dict(geom=L.GEOM_HOMOG, cost=L.COST_GROUPCORR)
    else:
        cam = synthetic.make_cameras(B, V, 4 * h, 4 * w)
        proj, dvals = mvsnet_cameras(cam["K"], cam["R"], cam["t"], cam["depth_min"], cam["depth_max"], D)
        cams = ops.proj_cams_device(proj.cuda().contiguous(), 0)
        dv = dvals[:, 0].contiguous().cuda()
        if case == "quad_per_pixel":
            dv = (dv.view(B, D, 1, 1) * (1.0 + 0.01 * torch.rand(B, 1, h, w, device="cuda"))).contiguous()
        kw = dict(cost=L.COST_VARIANCE_CVP if case == "generic_16ch" else L.COST_VARIANCE)
    L.set_tuning("warp_tiled", 0 if case == "quad_variance" else 4 if case == "lane_owner_variance" else 1)
    L.set_tuning("warp_gc_lds", 2 if case.startswith("homog_groupcorr_lds") else 0)      # (the LDS-staged group-correlation kernel, per-batch and per-pixel planes)
    try:
        full = ops.warp_cost(fcl[0], fcl[1:], cams, dv, out_dtype=torch.float16, **kw)
        for y0, hs_ in ((0, 20), (13, 25), (30, 22)):
            ref_slab = fcl[0][:, y0:y0 + hs_].contiguous()
            dv_slab = dv[:, :, y0:y0 + hs_].contiguous() if dv.dim() == 4 else dv
            slab = ops.warp_cost(ref_slab, fcl[1:], cams, dv_slab, out_dtype=torch.float16, ref_y0=y0, **kw)
            want = full[..., y0:y0 + hs_, :, :]
            assert slab.shape == want.shape
            assert torch.equal(slab, want), f"{case}: slab at row {y0} differs from the whole-image launch on {int((slab != want).sum())} values"
    finally:
        L.set_tuning("warp_tiled", -1)
        L.set_tuning("warp_gc_lds", 1)
