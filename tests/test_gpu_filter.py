"""GPU: pscv_geo_filter (one HIP launch) against the oracle and the reference's own masks.

The masks are thresholded fp32 quantities, so bit-exact agreement is required everywhere except on pixels whose
tested quantity lies within rounding distance of its threshold in the oracle (the kernel fuses multiply-adds that
ATen's 3-wide GEMMs round separately); those pixels are identified from the oracle's continuous quantities and must
be a small fraction of the image."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests._util import load_golden, t
from tests.test_oracle_filter import golden_inputs


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from wild_deep_mvs_amd import _lib as L, ops, synthetic
    from wild_deep_mvs_amd.evaluation import filtering as EF
    from oracle import filtering as OF
    L.lib()
    return L, ops, synthetic, EF, OF


def _compare(OF, ops, depth, src, K, R, tt, nc, p, label, rel=2e-4):
    want = OF.geometric_masks(depth, src, K, R, tt, max_reproj_error=p[0], depth_threshold=p[1], min_tri_angle=p[2], num_consistent=nc)
    q = OF.geometric_quantities(depth, src, K, R, tt)
    cams = ops.geo_filter_cams(K, R, tt).cuda()
    md, mp, mg, counts = ops.geo_filter(depth.cuda(), [s.cuda() for s in src], cams, max_reproj_error=p[0], depth_threshold=p[1],
                                        min_tri_angle=p[2], num_consistent=nc, want_counts=True)
    # per-(source, pixel) margins to the thresholds, relative
    zr, d = q["depth_reproj"], depth.expand_as(q["depth_reproj"])
    lim = torch.max(zr, d) * p[1]
    near = ((q["reproj_err"] - p[0]).abs() < rel * max(p[0], 1.0)) | (((zr - d).abs() - lim).abs() < rel * lim.abs().clamp(min=1e-6)) | \
           (zr.abs() < 1e-5) | (q["proj_depth_in_src"].abs() < 1e-5) | ((q["tri_angle"] - p[2]).abs() < rel * max(p[2], 1.0) + 2e-3)
    fragile = near.any(0)
    for name, got, key in (("mask_depth", md, "mask_depth"), ("mask_disp", mp, "mask_disp"), ("geo_mask", mg, "geo_mask")):
        diff = got.cpu() != want[key]
        assert int((diff & ~fragile).sum()) == 0, f"{label} {name}: {int((diff & ~fragile).sum())} mismatches away from any threshold"
        assert float(diff.float().mean()) < 2e-3, f"{label} {name}: {float(diff.float().mean()):.2e} of the pixels differ"
    for i, key in enumerate(("count_depth", "count_disp", "count_geo")):
        diff = counts[i].cpu() != want[key]
        assert int((diff & ~fragile).sum()) == 0, f"{label} {key}"
    # (samples that leave a source map read depth 0: the point collapses onto that camera centre, whose z in the
    #  reference frame is ~0 for this coplanar rig -- a legitimately fragile sign test on the image border)
    assert float(fragile.float().mean()) < 0.5, f"{label}: fragile set too large to be meaningful"
    return want, (md, mp, mg)


@pytest.mark.parametrize("fname", ["filter_tiny.npz", "filter_upsample.npz"])
def test_geo_filter_matches_reference_golden(env, fname):
    L, ops, synthetic, EF, OF = env
    g = load_golden(fname)
    depth, src, K, R, tt, nc = golden_inputs(g)
    p = [float(x) for x in g["params"]]
    want, got = _compare(OF, ops, depth, src, K, R, tt, nc, p, fname)
    # and directly against the reference's stored masks: at most a handful of threshold pixels
    for key, m in zip(("mask_depth", "mask_disp", "geo_mask"), got):
        ref = torch.from_numpy(np.asarray(g[key]).astype(bool))
        assert float((m.cpu() != ref).float().mean()) < 2e-3, key


@pytest.mark.parametrize("V,H,W,kw", [(9, 96, 128, dict(behind_view=5, half_res_view=2, near_view=7)),
                                      (3, 37, 53, dict()), (33, 40, 48, dict(near_view=1))])
def test_geo_filter_random_scenes(env, V, H, W, kw):
    """More views than the fixtures (up to the 32-source limit), ragged sizes, every special view at once."""
    L, ops, synthetic, EF, OF = env
    sc = synthetic.make_filter_scene(V, H, W, seed=V, **kw)
    for nc, p in ((3, (1.0, 0.01, 1.0)), (2, (0.5, 0.005, 3.0))):
        _compare(OF, ops, sc["depth"], sc["src_depth"], sc["K"], sc["R"], sc["t"], nc, p, f"V={V} {H}x{W} nc={nc}")


def test_geo_filter_identical_views_at_full_size(env):
    """Size-independent property at 1152x1600 with 8 sources: when every source IS the reference view (same camera,
    same depth map) the round trip is the identity -- reprojection error ~0 and relative depth difference ~0 pass for
    every pixel, the triangulation angle is 0 so geo_mask is empty."""
    L, ops, synthetic, EF, OF = env
    H, W, N = 1152, 1600, 8
    sc = synthetic.make_filter_scene(1, H, W, seed=1)
    K, R, tt = sc["K"].repeat(N + 1, 1, 1), sc["R"].repeat(N + 1, 1, 1), sc["t"].repeat(N + 1, 1, 1)
    d = sc["depth"].cuda()
    cams = ops.geo_filter_cams(K, R, tt).cuda()
    md, mp, mg, counts = ops.geo_filter(d, [d] * N, cams, num_consistent=N + 1, want_counts=True)
    # interior pixels: the bilinear sample at (x W/(W-1) - 0.5) mixes neighbours and is half outside on row/col 0
    assert float(mp[2:-2, 2:-2].float().mean()) > 0.999 and float(md[2:-2, 2:-2].float().mean()) > 0.999
    assert int(mg.sum()) == 0 and int(counts[2].max()) == 0


def test_filtering_run_mirror_writes_the_reference_files(env, tmp_path):
    """wild_deep_mvs_amd.evaluation.filtering.run: same files in, same files out as the reference's run()."""
    from argparse import Namespace
    L, ops, synthetic, EF, OF = env
    g = load_golden("filter_upsample.npz")
    meta = [int(x) for x in g["meta"]]
    V = meta[0]
    args = Namespace(model="m", nviews=V, data_path=str(tmp_path), scene="scene0", upsample=True, downscale=meta[7],
                     max_reproj_error=1.0, depth_threshold=0.01, min_tri_angle=1.0, num_consistent=meta[8], debug=False)
    folder = tmp_path / "IntRes" / "depthmaps" / EF.depth_folder_name(args) / "scene0"
    folder.mkdir(parents=True)
    np.savez(folder / "ref_out.npz", depthmap=g["depth"])
    for i in range(V - 1):
        np.savez(folder / f"src{i}_out.npz", depthmap=g[f"src_depth_{i}"])
    batch = {"filename": ["ref"], "K": t(g["K"]).unsqueeze(0), "R": t(g["R"]).unsqueeze(0), "t": t(g["t"]).unsqueeze(0),
             "src_filenames": [[f"src{i}"] for i in range(V - 1)]}
    EF.run([batch], args)
    out = np.load(tmp_path / "IntRes" / "geometric_filtering" / EF.depth_folder_name(args) / "scene0" / "ref_out.npz")
    for key in ("mask_depth", "mask_disp", "geo_mask"):
        assert out[key].dtype == np.bool_ and out[key].shape == np.asarray(g[key]).shape
        assert float((out[key] != np.asarray(g[key]).astype(bool)).mean()) < 2e-3, key
    assert (tmp_path / "IntRes" / "geometric_filtering" / EF.depth_folder_name(args) / "scene0" / "finished.txt").exists()
    EF.run([batch], args)     # second call: "already done", like the reference
