"""Multi-rank Vis-MVSNet source-view shard on ONE GPU box: two processes share cuda:0 and talk through gloo (RCCL needs
one GPU per rank; the collective calls, the partial-sum kernels and the gather of the pair results are the same code)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from _util import load_golden, t

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from wild_deep_mvs_amd import synthetic
        from wild_deep_mvs_amd.models.VisMVSNet.frontend import Frontend
        g = load_golden("vis_tiny.npz")
        H, W, V, seed, scene_seed = [int(x) for x in g["meta"][:5]]
        depth_nums = [int(x) for x in g["meta"][5:8]]
        scales = [float(x) for x in g["interval_scales"]]
        net = Frontend()
        net.load_state_dict(synthetic.sharpened_state_dict("vis", synthetic.template_of(net), seed=seed))
        net = net.cuda().eval()
        net.depth_nums, net.interval_scales = depth_nums, scales
        net.set_view_group(dist.group.WORLD)
        scene = {k: v.cuda() for k, v in synthetic.make_scene(1, V, H, W, seed=scene_seed).items()}
        out = net(scene["imgs"], scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"],
                  depth_nums=depth_nums, interval_scales=scales)
        q.put((rank, out["depth"].cpu().numpy(), [[p[0].cpu().numpy() for p in st] for st in out["depth_pair_list"]]))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_vis_source_view_shard_two_ranks_one_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = load_golden("vis_tiny.npz")
    ref = g["depth"]
    for rank, depth, pairs in res:
        rel = np.abs(depth - ref).mean() / np.abs(ref).mean()
        print(f"[parity] view-shard rank {rank}: depth rel-L1 vs reference {rel:.3e}", flush=True)
        assert rel <= 1e-3
        # pair results of ALL views on every rank, in view order
        for si, st in enumerate(pairs):
            assert len(st) == 2
            for vi, ed in enumerate(st):
                r = g[f"pair_depth_s{3 - si}_v{vi}"]
                assert np.abs(ed - r).mean() / np.abs(r).mean() <= 2e-3
    assert np.abs(res[0][1] - res[1][1]).max() <= 1e-5, "ranks must agree on the fused result"
