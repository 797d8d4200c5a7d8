"""Overlap soak: every kernel family of the engine, launched on one HIP stream while the regulariser's 32->8 depth sweep
(conv0 at the headline size: LDS ring + MFMA, the partner that exposed the defect of DESIGN.md section 6) runs on a second
stream, must store the SAME BITS as its solo launch -- `include/pscv.h` promises "re-entrant per stream".  >= 200 launches per
family at sizes of ~50 us and more, through the C ABI with DEFAULT tuning.  (Round 3 found the packed-fp32 build of the LDS-staged
warp kernel wrong under exactly this overlap; the kernel ships as its scalar build.  The last test runs the stand-alone reproducer
of the instruction-level cause and REPORTS what this box shows.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LAUNCHES = 200
BATCH = 8


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from wild_deep_mvs_amd import _lib as L, ops, synthetic
    return L, ops, synthetic


class Soak:
    """Partner = conv0 (32 -> 8 sweep over 192 x 128 x 160) on stream B, repeated so that it covers the victim's launches."""

    def __init__(self, L, ops):
        self.L, self.ops = L, ops
        g = torch.Generator().manual_seed(1)
        self.px = (torch.randn(1, 192, 128, 160, 32, generator=g) * 0.5).to(torch.float16).cuda()
        w = torch.randn(8, 32, 3, 3, 3, generator=g) / np.sqrt(27 * 32)
        self.player = ops.Conv3dLayer.build(w, kind=L.CONV_S1, device="cuda", relu=True, dtype=torch.float16)
        self.sa, self.sb = torch.cuda.Stream(), torch.cuda.Stream()
        self.partner_us = self._time(lambda: ops.conv3d(self.px, self.player))

    @staticmethod
    def _time(fn, n=5):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    def run(self, name, victim, *, launches=LAUNCHES, rel_tol=None):
        """victim() -> tuple of tensors.  Returns (bad launches, worst relative deviation, overlap ratio)."""
        as_tuple = lambda o: tuple(o) if isinstance(o, (tuple, list)) else (o,)
        solo = tuple(x.clone() for x in as_tuple(victim()))
        torch.cuda.synchronize()
        vic_us = self._time(lambda: victim())
        per = max(1, int(round(vic_us / self.partner_us + 0.5)))          # partner launches per victim launch
        bad, worst, t_overlap = 0, 0.0, 0.0
        cur = torch.cuda.current_stream()
        for it in range(0, launches, BATCH):
            self.sa.wait_stream(cur); self.sb.wait_stream(cur)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
            self.sa.wait_event(e0); self.sb.wait_event(e0)
            outs = []
            for _ in range(BATCH):
                with torch.cuda.stream(self.sb):
                    for _ in range(per):
                        self.ops.conv3d(self.px, self.player)
                with torch.cuda.stream(self.sa):
                    outs.append(as_tuple(victim()))
            cur.wait_stream(self.sa); cur.wait_stream(self.sb)
            e1.record(cur)
            torch.cuda.synchronize()
            t_overlap += e0.elapsed_time(e1) * 1e3
            for o in outs:
                same = all(torch.equal(a, b) for a, b in zip(o, solo))
                if not same:
                    dev = max(float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-30)) for a, b in zip(o, solo))
                    worst = max(worst, dev)
                    if rel_tol is None or not dev <= rel_tol:
                        bad += 1
        n = (launches + BATCH - 1) // BATCH * BATCH
        serial = n * (vic_us + per * self.partner_us)
        print(f"[overlap] {name}: victim {vic_us:.0f} us, partner {per} x {self.partner_us:.0f} us; {bad} of {n} launches differ from the solo launch"
              f" (worst rel {worst:.2e}); both streams took {t_overlap / serial:.2f} of their serial time", flush=True)
        return bad, worst, t_overlap / serial


@pytest.fixture(scope="module")
def soak(env):
    L, ops, synthetic = env
    return Soak(L, ops)


def _warp_inputs(ops, synthetic, V, C, h, w, D, dtype, seed=3):
    from wild_deep_mvs_amd.models.MVSNet.model import build_proj_matrices
    cam = synthetic.make_cameras(1, V, 4 * h, 4 * w)
    K = cam["K"].clone(); K[:, :, :2] /= 4
    proj = build_proj_matrices(K, cam["R"], cam["t"])
    cams = ops.proj_cams_device(proj.cuda().float().contiguous(), 0)
    steps = torch.arange(D, dtype=torch.float32).view(1, -1)
    dv = (cam["depth_min"][:, :1] + (cam["depth_max"][:, :1] - cam["depth_min"][:, :1]) / (D - 1) * steps).contiguous().cuda()
    feats = synthetic.make_features(1, V, C, h, w, seed=seed)
    fcl = [ops.to_channels_last(feats[i].cuda(), dtype) for i in range(V)]
    return fcl, cams, dv


WARP_CASES = [("lds variance", 1, "variance", 32, False), ("lds softmin", 1, "softmin", 32, False), ("lane-owner variance", 4, "variance", 32, False),
              ("quad variance", 0, "variance", 32, False),
              ("quad per-pixel planes", 1, "variance", 32, True), ("generic 16 channels", 1, "variance_cvp", 16, False)]


@pytest.mark.parametrize("name,tiled,cost,C,per_pixel", WARP_CASES, ids=[c[0].replace(" ", "_") for c in WARP_CASES])
def test_warp_cost_kernels_are_bit_stable_beside_conv0(env, soak, name, tiled, cost, C, per_pixel):
    L, ops, synthetic = env
    fcl, cams, dv = _warp_inputs(ops, synthetic, 5, C, 128, 160, 192, torch.float16)
    if per_pixel:
        dv = (dv.view(1, -1, 1, 1) + 0.01 * torch.rand(1, 1, 128, 160, device="cuda")).contiguous()
    code = {"variance": L.COST_VARIANCE, "softmin": L.COST_SOFTMIN, "variance_cvp": L.COST_VARIANCE_CVP}[cost]
    assert L.get_tuning("warp_tiled") == 1
    L.set_tuning("warp_tiled", tiled)
    try:
        bad, _, _ = soak.run(f"warp_cost {name}", lambda: ops.warp_cost(fcl[0], fcl[1:], cams, dv, cost=code, temp=0.7, out_dtype=torch.float16))
    finally:
        L.set_tuning("warp_tiled", -1)
    assert bad == 0


@pytest.mark.parametrize("kernel", ["lds", "quad"])
def test_groupcorr_homog_warp_is_bit_stable_beside_conv0(env, soak, kernel):
    """Vis-MVSNet's sweep: HOMOG geometry, group-wise correlation, 4 source views, per-batch planes: the LDS-staged kernel
    (warp_gc_lv.hip, the default for these planes) and the quad kernel (per-pixel planes, `warp_gc_lds` = 0)."""
    L, ops, synthetic = env
    from oracle import vismvsnet as OV
    V, H, W, D = 5, 512, 640, 64
    sc = synthetic.make_scene(1, V, H, W, seed=5)
    di = (sc["depth_max"] - sc["depth_min"]) / 128
    cams = [OV.fill_cam_array(sc["K"][:, i], sc["R"][:, i], sc["t"][:, i], sc["depth_min"][:, i], di[:, i]) for i in range(V)]
    blocks = ops.homog_cams_device(cams[0].cuda(), [c.cuda() for c in cams[1:]], 0.5)
    feats = synthetic.make_features(1, V, 32, H // 2, W // 2, seed=4)
    fcl = [ops.to_channels_last(feats[i].cuda(), torch.float16) for i in range(V)]
    planes = (cams[0][:, 1, 3, 0].view(1, 1) + di[:, :1] * 2.0 * torch.arange(D, dtype=torch.float32).view(1, D)).contiguous().cuda()
    assert L.get_tuning("warp_gc_lds") == 1
    L.set_tuning("warp_gc_lds", 1 if kernel == "lds" else 0)
    try:
        bad, _, _ = soak.run(f"warp_cost groupcorr HOMOG ({kernel})", lambda: ops.warp_cost(fcl[0], fcl[1:], blocks, planes, geom=L.GEOM_HOMOG,
                                                                                          cost=L.COST_GROUPCORR, out_dtype=torch.float16))
    finally:
        L.set_tuning("warp_gc_lds", 1)
    assert bad == 0


# (name, c_in, c_out, kind, input D,H,W, with skip, tuning) -- one entry per conv3d kernel of csrc/
CONV3D_CASES = [
    ("sweep8 32->8", 32, 8, "S1", (192, 128, 160), False, {}),
    ("sweep8_kdm 32->8", 32, 8, "S1", (192, 128, 160), False, {"sweep_kdm": 1}),
    ("sweep_s2 8->16", 8, 16, "S2", (192, 256, 320), False, {}),
    ("sweep_s2 8->32 (Vis block + shortcut)", 8, 32, "S2", (128, 256, 320), False, {}),
    ("brick S1 16->16", 16, 16, "S1", (192, 128, 160), False, {}),
    ("brick S2 16->32", 16, 32, "S2", (192, 128, 160), False, {}),
    ("brick S1 32->32", 32, 32, "S1", (128, 64, 80), False, {}),
    ("brick S2 32->64", 32, 64, "S2", (192, 64, 80), False, {}),
    ("brick S1 64->64", 64, 64, "S1", (48, 64, 80), False, {}),
    ("brick T2 64->32", 64, 32, "T2", (48, 48, 40), True, {}),
    ("brick T2 32->16", 32, 16, "T2", (96, 64, 40), True, {}),
    ("t2p8 16->8", 16, 8, "T2", (96, 128, 160), True, {}),
    ("c1_sweep 8->1", 8, 1, "S1", (192, 256, 160), False, {}),
    ("c1 8->1", 8, 1, "S1", (192, 256, 160), False, {"c1_sweep": 0}),
    ("sweepc 8->8", 8, 8, "S1", (64, 256, 320), True, {}),
    ("sweepc 16->8", 16, 8, "S1", (64, 256, 320), False, {}),
]


@pytest.mark.parametrize("name,cin,cout,kind,dhw,with_skip,tune", CONV3D_CASES, ids=[c[0].split(" (")[0].replace(" ", "_").replace("->", "to") for c in CONV3D_CASES])
def test_conv3d_kernels_are_bit_stable_beside_conv0(env, soak, name, cin, cout, kind, dhw, with_skip, tune):
    L, ops, synthetic = env
    g = torch.Generator().manual_seed(cin * 100 + cout)
    kcode = {"S1": L.CONV_S1, "S2": L.CONV_S2, "T2": L.CONV_T2}[kind]
    tr = kind == "T2"
    w = torch.randn((cin, cout, 3, 3, 3) if tr else (cout, cin, 3, 3, 3), generator=g) / np.sqrt(27 * cin)
    layer = ops.Conv3dLayer.build(w, kind=kcode, transposed=tr, device="cuda", relu=cout > 1, dtype=torch.float16,
                                  conv_bias=torch.zeros(1) if cout == 1 else None)
    x = (torch.randn(1, *dhw, cin, generator=g) * 0.5).to(torch.float16).cuda()
    skip = None
    if with_skip:
        so = tuple(2 * v for v in dhw) if tr else dhw
        skip = (torch.randn(1, *so, cout, generator=g) * 0.5).to(torch.float16).cuda()
    for k, v in tune.items():
        L.set_tuning(k, v)
    try:
        bad, _, _ = soak.run(f"conv3d {name}", lambda: ops.conv3d(x, layer, skip=skip))
    finally:
        for k in tune:
            L.set_tuning(k, 1 if k == "c1_sweep" else 0)
    assert bad == 0


def test_block8_and_cat2_sweeps_are_bit_stable_beside_conv0(env, soak):
    L, ops, synthetic = env
    g = torch.Generator().manual_seed(9)
    mk = lambda ci, relu: ops.Conv3dLayer.build(torch.randn(8, ci, 3, 3, 3, generator=g) / np.sqrt(27 * ci), kind=L.CONV_S1, device="cuda",
                                               relu=relu, relu_post=not relu, dtype=torch.float16)
    l1, l2, l16 = mk(8, True), mk(8, False), mk(16, True)
    x = (torch.randn(1, 64, 256, 320, 8, generator=g) * 0.5).to(torch.float16).cuda()
    x2 = (torch.randn(1, 64, 256, 320, 8, generator=g) * 0.5).to(torch.float16).cuda()
    assert ops.conv3d_block8(x, l1, l2) is not None, "the fused BasicBlock launch applies to 8 -> 8 sweep layers"
    bad, _, _ = soak.run("conv3d_block8 (two 8->8 sweeps + residual)", lambda: ops.conv3d_block8(x, l1, l2))
    assert bad == 0
    bad, _, _ = soak.run("conv3d_cat2 16->8 (two 8-channel inputs)", lambda: ops.conv3d(x, l16, x2=x2))
    assert bad == 0


CONV2D_CASES = [("k3 s1 8->8", 8, 8, 3, 1, (15, 512, 640)), ("k5 s2 8->16", 8, 16, 5, 2, (20, 512, 640)), ("k3 s1 32->32", 32, 32, 3, 1, (20, 128, 160)),
                ("wlds k3 s1 64->64", 64, 64, 3, 1, (1, 1024, 1280))]


@pytest.mark.parametrize("name,cin,cout,ks,stride,bhw", CONV2D_CASES, ids=[c[0].replace(" ", "_").replace("->", "to") for c in CONV2D_CASES])
def test_conv2d_kernels_are_bit_stable_beside_conv0(env, soak, name, cin, cout, ks, stride, bhw):
    L, ops, synthetic = env
    g = torch.Generator().manual_seed(cin + cout + ks)
    w = torch.randn(cout, cin, ks, ks, generator=g) / np.sqrt(ks * ks * cin)
    layer = ops.Conv2dLayer.build(w, stride=stride, device="cuda", relu=True, dtype=torch.float16)
    x = (torch.randn(*bhw, cin, generator=g) * 0.5).to(torch.float16).cuda()
    bad, _, _ = soak.run(f"conv2d {name}", lambda: ops.conv2d(x, layer))
    assert bad == 0


def test_tail_and_vis_glue_kernels_are_bit_stable_beside_conv0(env, soak):
    """softargmin (all outputs), fuse_pairs, uncert_net."""
    L, ops, synthetic = env
    g = torch.Generator().manual_seed(2)
    logits = torch.randn(4, 192, 128, 160, generator=g).cuda()
    dv = torch.linspace(2, 6, 192).view(1, -1).repeat(4, 1).cuda()

    def sam():
        o = ops.softargmin(logits, dv, want_index=True, want_conf=True, want_entropy=True)
        return tuple(o[k] for k in sorted(o))
    assert soak.run("softargmin", sam)[0] == 0
    vols = [(torch.randn(1, 64, 256, 320, 8, generator=g) * 0.5).to(torch.float16).cuda() for _ in range(4)]
    unc = [torch.randn(1, 256, 320, generator=g).cuda() for _ in range(4)]
    assert soak.run("fuse_pairs", lambda: ops.fuse_pairs(vols, unc))[0] == 0
    prm = ops.pack_uncert_params(torch.randn(8, 1, 3, 3, generator=g), tuple(torch.rand(8, generator=g) + 0.5 for _ in range(4)),
                                 torch.randn(8, 8, 3, 3, generator=g) / 8, tuple(torch.rand(8, generator=g) + 0.5 for _ in range(4)),
                                 torch.randn(1, 8, 3, 3, generator=g) / 8).cuda()
    ent = torch.rand(8, 576, 800, generator=g).cuda()
    assert soak.run("uncert_net", lambda: ops.uncert_net(ent, prm))[0] == 0


def test_backward_kernels_are_stable_beside_conv0(env, soak):
    """Weight gradient (fixed-order reduction: bit-stable) and the warp backward (its flush adds with global float atomics in
    arrival order, so solo launches already differ in the last bits: held to 1e-5 of the gradient's range instead)."""
    L, ops, synthetic = env
    g = torch.Generator().manual_seed(6)
    p = (torch.randn(1, 96, 128, 160, 8, generator=g) * 0.5).to(torch.float16).cuda()
    q = (torch.randn(1, 96, 128, 160, 32, generator=g) * 0.5).to(torch.float16).cuda()
    assert soak.run("conv3d_wgrad 8x32", lambda: ops.conv3d_wgrad(p, q, ca=8, cb=32, stride=1))[0] == 0
    q2 = q.repeat(2, 1, 1, 1, 1)
    assert soak.run("bn_stats", lambda: ops.bn_stats(q2))[0] == 0
    fcl, cams, dv = _warp_inputs(ops, synthetic, 5, 32, 128, 160, 48, torch.float16)
    go = (torch.randn(1, 48, 128, 160, 32, generator=g) * 0.1).to(torch.float16).cuda()

    def wb():
        dref, dsrcs, _ = ops.warp_cost_bwd(fcl[0], fcl[1:], cams, dv, go, cost=L.COST_VARIANCE)
        return (dref,) + tuple(dsrcs)
    assert soak.run("warp_cost_bwd variance", wb, launches=96, rel_tol=1e-5)[0] == 0


def test_report_the_platform_defect_with_the_minimal_reproducer(env, soak):
    """Diagnostic, never fails: the LDS-free, self-checking victims of scripts/ubench/lds_pk_overlap.hip (`opsel_victim`: packed fp32
    instructions with one op_sel bit set, each result checked in the kernel against plain v_mul / v_add / v_fma) beside conv0.  Round 4
    measured: millions of wrong LOW results in lanes 48-63 for the forms whose op_sel bit of SRC1 is set (v_pk_mul_f32 op_sel:[0,1],
    v_pk_add_f32 op_sel:[0,1], v_pk_fma_f32 op_sel:[0,1,0]); 0 for src0 / src2 selectors, op_sel_hi forms, v_pk_mov_b32; 0 for every
    form when launched alone.  scripts/lint_isa.py keeps the failing forms out of libpscv.so."""
    import ctypes as C
    import os
    L, ops, synthetic = env
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "ubench", "liblpo.so")
    if not os.path.exists(path):
        pytest.skip("scripts/ubench/liblpo.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    lpo = C.CDLL(path)
    lpo.lpo_opsel_victim.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    names = ["v_pk_mov_b32 op_sel:[1,0]", "v_pk_mul_f32 op_sel:[1,0]", "v_pk_mul_f32 op_sel:[0,1]", "v_pk_add_f32 op_sel:[1,0]", "v_pk_add_f32 op_sel:[0,1]",
             "v_pk_fma_f32 op_sel:[1,0,0]", "v_pk_fma_f32 op_sel:[0,1,0]", "v_pk_fma_f32 op_sel:[0,0,1]", "v_pk_fma_f32 op_sel_hi:[1,0,1]"]
    for partner in (False, True):
        errs = torch.zeros(36, dtype=torch.int32, device="cuda")
        for it in range(40):
            if partner:
                with torch.cuda.stream(soak.sb):
                    for _ in range(3):
                        ops.conv3d(soak.px, soak.player)
            with torch.cuda.stream(soak.sa):
                assert lpo.lpo_opsel_victim(errs.data_ptr(), 400, soak.sa.cuda_stream) == 0
            torch.cuda.synchronize()
        e = errs.cpu().tolist()
        print(f"[op_sel reproducer] conv0 on a second stream: {partner}; wrong results per 16-lane group (lanes 0-15 / 16-31 / 32-47 / 48-63), "
              "40 launches x 1 M threads x 400 rounds:", flush=True)
        for k, nm in enumerate(names):
            print(f"[op_sel reproducer]    {nm:32s} {e[4 * k:4 * k + 4]}", flush=True)


def test_post_path_and_loss_kernels_are_bit_stable_beside_conv0(env, soak):
    """The kernels in which the ISA lint found the unreliable packed form in round 3's build (geo_filter, photo_warp, cvp_cams: now
    compiled with -fno-slp-vectorize), plus their neighbours: SSIM forward, the photometric warp's backward, calDepthHypo."""
    L, ops, synthetic = env
    # geometric filter: 1152 x 1600, ten source views (evaluation/filtering.py:60-85)
    sc = synthetic.make_filter_scene(11, 1152, 1600, seed=2)
    cams = ops.geo_filter_cams(sc["K"], sc["R"], sc["t"]).cuda()
    depth, src = sc["depth"].cuda(), [d.cuda() for d in sc["src_depth"]]
    assert soak.run("geo_filter 1152x1600 x 10 views", lambda: ops.geo_filter(depth, src, cams, want_counts=True))[0] == 0
    # photometric warp + its backward + SSIM (models/trainer.py:209-278, utils/ssimLoss.py)
    pc = synthetic.make_photo_case(2, 5, 512, 640, seed=3)
    from wild_deep_mvs_amd.models.MVSNet.model import build_proj_matrices
    proj = build_proj_matrices(pc["K"], pc["R"], pc["t"]).cuda().float()
    inv_ref = ops.inv_proj4x4(proj[:, 0]).contiguous()
    srcs = pc["imgs"][:, 1:].cuda().contiguous()
    d0 = pc["depths"][0].cuda().contiguous()
    pw = lambda: tuple(v for v in ops.photo_warp(srcs, d0, inv_ref, proj[:, 1:].contiguous(), want_z=True, want_flows=True).values() if v is not None)
    assert soak.run("photo_warp 2 x 4 sources 512x640", pw)[0] == 0
    gw = torch.rand(2, 4, 3, 512, 640, device="cuda")
    assert soak.run("photo_warp_bwd", lambda: ops.photo_warp_bwd(srcs, d0, inv_ref, proj[:, 1:].contiguous(), gw))[0] == 0
    a, b = torch.rand(2, 3, 512, 640, device="cuda"), torch.rand(8, 3, 512, 640, device="cuda")
    assert soak.run("ssim 8 x 3 x 512x640", lambda: ops.ssim(a, b))[0] == 0
    # CVP: camera blocks of all levels in one launch + calDepthHypo at 1024 x 1280
    B, V, H, W = 1, 5, 1024, 1280
    cs = synthetic.make_scene(B, V, 64, 80, seed=4)            # (cameras only; the images are not used)
    row = torch.tensor([0., 0., 0., 1.])
    ref_ex = torch.cat((torch.cat((cs["R"][:, 0], cs["t"][:, 0] * 8), 2), row.view(1, 1, 4).expand(B, 1, 4)), 1).cuda()
    src_ex = torch.cat((torch.cat((cs["R"][:, 1:], cs["t"][:, 1:] * 8), 3), row.view(1, 1, 1, 4).expand(B, V - 1, 1, 4)), 2).cuda()
    K = cs["K"].cuda() * torch.tensor([16.0, 16.0, 1.0], device="cuda").view(1, 1, 3, 1)
    cc = lambda: ops.cvp_cams(K[:, 0], K[:, 1:], ref_ex, src_ex, [1.0, 2.0, 4.0, 8.0, 16.0])
    assert soak.run("cvp_cams 5 levels", cc, launches=96)[0] == 0
    _, hypo_cams = ops.cvp_cams(K[:, 0], K[:, 1:], ref_ex, src_ex, [1.0, 2.0, 4.0, 8.0, 16.0])
    dmap = (2.5 + 3.0 * torch.rand(B, H, W)).cuda()
    fb = torch.full((B,), 4.0 / 128, device="cuda")
    assert soak.run("cvp_depth_hypos 1024x1280", lambda: ops.cvp_depth_hypos(dmap, hypo_cams[0], fb))[0] == 0
