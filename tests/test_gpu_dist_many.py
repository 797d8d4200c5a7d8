"""Every sharded path of the engine at world = 4 and world = 8: the ranks are processes that SHARE cuda:0 and talk through gloo
(RCCL needs one GPU per rank; the kernels, the slab / halo / merge logic and the collective call sites are the same code -- only
the transport differs, `wild_deep_mvs_amd.dist.exchange`).  Reduced sizes, chosen so that at eight ranks every rank still owns
work in every sharded stage and the reduce-scatter form of the Vis fusion is taken (slabs >= the 8-unit halo).

Asserted: sharded == unsharded (same process, same inputs) within the bars of the two-rank tests, on every rank, and the ranks
agree with each other.  Printed for the log: the Vis source-view shard's depth error at eight ranks with 16-bit shares (the
default, SURVEY 8e's payload budget) and with fp32 shares (`SingleStage.view_reduce_fp32`): what the summation order of a 16-bit
reduction costs at that world size.  Also here: the CVP row-slab shard of the refinement levels (two and four ranks)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from _util import retry_infra as _retry_infra

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _rel(a, b):
    return float((a - b).abs().mean() / b.abs().mean())


def _mvsnet_depth(dev, world):
    from wild_deep_mvs_amd import synthetic
    from wild_deep_mvs_amd.models.MVSNet.model import MVSNet
    net = MVSNet("variance")
    net.load_state_dict(synthetic.sharpened_state_dict("mvsnet", synthetic.template_of(net), seed=0))
    net = net.to(dev).eval()
    net.num_depth = 64                      # 8 ranks x 8 planes: the smallest block the U-Net's three stride-2 levels allow
    sc = {k: v.to(dev) for k, v in synthetic.make_scene(1, 3, 128, 160, seed=4).items()}
    call = lambda: net(sc["imgs"], sc["K"], sc["R"], sc["t"], sc["depth_min"], sc["depth_max"])
    with torch.no_grad():
        ref = call()
        net.set_depth_group(dist.group.WORLD)
        got = call()
        net.set_depth_group(None)
    return {"depth": _rel(got["depth"], ref["depth"]),
            "conf": float((got["photometric_confidence"] - ref["photometric_confidence"]).abs().mean())}, got["depth"]


def _vis(dev, world, mode):
    from wild_deep_mvs_amd import synthetic
    from wild_deep_mvs_amd.models.VisMVSNet.frontend import Frontend
    net = Frontend()
    net.load_state_dict(synthetic.sharpened_state_dict("vis", synthetic.template_of(net), seed=0))
    net = net.to(dev).eval()
    if mode == "view":
        # 8 source views; stage volumes (64,16,20), (8,32,40), (8,64,80): at 8 ranks stage 1 cuts 8-plane depth slabs and stage 3
        # 8-row slabs (reduce-scatter + halo exchange), stage 2 is thinner than the halo -> 16-bit all-reduce + replicated RegFuse
        V, H, W, kw = 9, 128, 160, dict(depth_nums=[64, 8, 8], interval_scales=[1.0, 1.0, 0.5])
    else:
        # stage heights 32 / 64 / 128; every rank owns >= 2 planes of every stage at 8 ranks ("depth") resp. >= 8 rows of stages
        # 2-3 ("depth_rows"); 3 views < 8 ranks: the view-sharded 2-D extractor must fall back to the replicated one on ALL ranks
        V, H, W, kw = 3, 256, 320, dict(depth_nums=[64, 16, 16], interval_scales=[2.0, 2.0, 1.0])
    net.depth_nums, net.interval_scales = kw["depth_nums"], kw["interval_scales"]
    sc = {k: v.to(dev) for k, v in synthetic.make_scene(1, V, H, W, seed=6).items()}
    call = lambda: net(sc["imgs"], sc["K"], sc["R"], sc["t"], sc["depth_min"], sc["depth_max"], **kw)
    setter = {"depth": net.set_depth_group, "depth_rows": net.set_depth_row_groups, "view": net.set_view_group}[mode]
    rec = {}
    with torch.no_grad():
        ref = call()
        setter(dist.group.WORLD)
        got = call()
        rec["depth"] = _rel(got["depth"], ref["depth"])
        rec["stages"] = [_rel(a, b) for a, b in zip(got["depth_est_list"], ref["depth_est_list"])]
        rec["prob"] = float((got["photometric_confidence"] - ref["photometric_confidence"]).abs().mean())
        if mode == "view":
            for st in (net.model.stage1, net.model.stage2, net.model.stage3):
                st.view_reduce_fp32 = True
            got32 = call()
            for st in (net.model.stage1, net.model.stage2, net.model.stage3):
                st.view_reduce_fp32 = False
            rec["depth_fp32_shares"] = _rel(got32["depth"], ref["depth"])
            rec["stages_fp32_shares"] = [_rel(a, b) for a, b in zip(got32["depth_est_list"], ref["depth_est_list"])]
        setter(None)
    return rec, got["depth"]


def _cvp_rows(dev, world):
    from wild_deep_mvs_amd import synthetic
    from wild_deep_mvs_amd.models.CVP_MVSNet.frontend import Frontend
    net = Frontend()
    net.load_state_dict(synthetic.sharpened_state_dict("cvp", synthetic.template_of(net), seed=0))
    net = net.to(dev).eval()
    sc = {k: v.to(dev) for k, v in synthetic.make_scene(1, 3, 256, 320, seed=7).items()}      # levels 64x80 (coarse), 128x160, 256x320
    call = lambda: net(sc["imgs"], sc["K"], sc["R"], sc["t"], sc["depth_min"], sc["depth_max"], nscale=3)
    with torch.no_grad():
        ref = call()
        net.set_row_group(dist.group.WORLD)
        got = call()
        net.set_row_group(None)
    return {"depth": _rel(got["depth"], ref["depth"]),
            "levels": [_rel(a, b) for a, b in zip(got["depth_est_list"], ref["depth_est_list"])],
            "levels_max": [float((a - b).abs().max() / b.abs().max()) for a, b in zip(got["depth_est_list"], ref["depth_est_list"])],
            "conf": float((got["photometric_confidence"] - ref["photometric_confidence"]).abs().mean())}, got["depth"]


def _worker(rank, world, port, q, modes):
    _init(rank, world, port)
    try:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        out = {}
        for mode in modes:
            if mode == "mvsnet_depth":
                rec, depth = _mvsnet_depth(dev, world)
            elif mode == "cvp_rows":
                rec, depth = _cvp_rows(dev, world)
            else:
                rec, depth = _vis(dev, world, mode)
            # the ranks must hold the SAME final map (every collective ends in an all-gather / merged statistics)
            mine = depth.float().contiguous()
            ref0 = mine.clone()
            dist.broadcast(ref0, src=0)
            rec["rank_disagreement"] = float((mine - ref0).abs().max())
            out[mode] = rec
            torch.cuda.empty_cache()
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def _run(world, modes, timeout=420):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, modes)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=timeout) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0, f"a rank process exited with code {p.exitcode}"
    return res


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [4, 8])
@_retry_infra
def test_every_sharded_path_equals_the_unsharded_forward_at_4_and_8_ranks(world):
    """MVSNet depth planes (per-layer halo exchange + LSE merge), Vis depth planes, Vis planes + rows + view-sharded extractor
    (world > V: the replicated extractor on every rank, decided before any collective), Vis source views (reduce-scatter into
    slabs + halo exchange / 16-bit all-reduce).  Bars: final depth <= 1e-3 of the unsharded forward (north-star bar; the two-rank
    tests hold 2-5e-4), fp32 shares <= 3e-4, ranks agree to the last bit."""
    res = _run(world, ["mvsnet_depth", "depth", "depth_rows", "view"])
    for rank, out in res:
        for mode, rec in out.items():
            print(f"[parity] world {world} rank {rank} {mode}: " + ", ".join(f"{k}={v if not isinstance(v, float) else format(v, '.2e')}" for k, v in rec.items()), flush=True)
            assert rec["rank_disagreement"] == 0.0, (mode, rec)
            assert rec["depth"] <= 1e-3, (mode, rec)
        assert out["mvsnet_depth"]["depth"] <= 2e-4 and out["depth"]["depth"] <= 3e-4 and out["depth_rows"]["depth"] <= 3e-4, out
        assert out["view"]["depth_fp32_shares"] <= 3e-4, out["view"]
    v = res[0][1]["view"]
    print(f"[parity] Vis source-view shard at {world} ranks, final depth rel-L1 vs unsharded: 16-bit shares {v['depth']:.2e} "
          f"(stages {', '.join(format(x, '.1e') for x in v['stages'])}), fp32 shares {v['depth_fp32_shares']:.2e} "
          f"(stages {', '.join(format(x, '.1e') for x in v['stages_fp32_shares'])})", flush=True)


@pytest.mark.timeout(400)
@pytest.mark.parametrize("world", [2, 4])
@_retry_infra
def test_cvp_row_slab_shard_of_the_refinement_levels(world):
    """CVP-MVSNet (reference models/CVP_MVSNet/models/net.py:166-219) with the image rows of the two refinement levels (128x160,
    256x320; 8 per-pixel planes) sharded: slab + 20-row recomputed halo (the regulariser reaches +-17), whole-image cameras with the
    slab's row origin (`pscv_warp_cost_rows`), one all-gather of the owned rows per level.  The coarsest level runs replicated.
    Level depths against the unsharded forward: a smaller volume may select another conv kernel variant (another fp32 summation
    order in front of a 16-bit rounding), so rel-L1 <= 2e-4 per level like the Vis row shard; ranks agree exactly."""
    res = _run(world, ["cvp_rows"])
    for rank, out in res:
        rec = out["cvp_rows"]
        print(f"[parity] CVP row-slab shard world {world} rank {rank}: final depth rel-L1 {rec['depth']:.2e}, levels (fine -> coarse) "
              + ", ".join(f"{a:.1e} (max {b:.1e})" for a, b in zip(rec["levels"], rec["levels_max"])) + f", confidence mean abs {rec['conf']:.1e}", flush=True)
        assert rec["rank_disagreement"] == 0.0
        assert rec["levels"][-1] == 0.0, "the coarsest level is replicated: identical"
        assert all(e <= 2e-4 for e in rec["levels"]) and rec["conf"] <= 5e-3, rec
