"""World-size-2 gloo tests (CPU) of the multi-GPU sharding rules in wild_deep_mvs_amd/dist.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from wild_deep_mvs_amd import dist as pdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _partials(logits, depth, index_offset):
    """torch restatement of pscv_softargmin's out_partials for one depth shard: [B,4,h,w]."""
    m = logits.max(dim=1).values
    e = torch.exp(logits - m.unsqueeze(1))
    idx = torch.arange(logits.shape[1], dtype=torch.float32).view(1, -1, 1, 1) + index_offset
    return torch.stack([m, e.sum(1), (e * depth.view(1, -1, 1, 1)).sum(1), (e * idx).sum(1)], dim=1)


def _worker(rank, world, port, D, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)             # same full problem on every rank
        logits = torch.randn(2, D, 6, 7, generator=g) * 4
        depth = torch.linspace(2.0, 6.0, D)
        d0, d1 = pdist.plane_shard(D, world, rank, multiple=8)
        part = _partials(logits[:, d0:d1], depth[d0:d1], d0)
        got_depth, got_index = pdist.merge_partials(part)
        p = torch.softmax(logits, 1)
        want_depth = (p * depth.view(1, -1, 1, 1)).sum(1)
        want_index = (p * torch.arange(D, dtype=torch.float32).view(1, -1, 1, 1)).sum(1)
        out_q.put((rank, float((got_depth - want_depth).abs().max()), float((got_index - want_index).abs().max()),
                   (d0, d1), pdist.view_shard(5, world, rank)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_depth_plane_shard_lse_merge_world2():
    world, D = 2, 48
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, D, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert [r[3] for r in res] == [(0, 24), (24, 48)]
    assert [r[4] for r in res] == [[0, 2, 4], [1, 3]]
    for _, ed, ei, _, _ in res:
        assert ed < 1e-5 and ei < 1e-4


def test_plane_shard_covers_range_with_aligned_boundaries():
    for D, world, mult in [(192, 8, 8), (192, 5, 8), (48, 4, 8), (64, 3, 2), (8, 8, 8)]:
        spans = [pdist.plane_shard(D, world, r, mult) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == D
        for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
            assert a1 == b0
        assert all(a % mult == 0 and b % mult == 0 and b >= a for a, b in spans)
    with pytest.raises(ValueError):
        pdist.plane_shard(50, 2, 0, 8)


def test_merge_is_exact_for_unequal_maxima():
    torch.manual_seed(1)
    logits = torch.randn(1, 32, 3, 3) * 10
    logits[:, 20:] += 50.0                                # second shard dominates by e^50
    depth = torch.linspace(1.0, 2.0, 32)
    parts = [_partials(logits[:, :16], depth[:16], 0), _partials(logits[:, 16:], depth[16:], 16)]
    d, i = pdist.merge_partials_local(parts)
    p = torch.softmax(logits, 1)
    np.testing.assert_allclose(d.numpy(), (p * depth.view(1, -1, 1, 1)).sum(1).numpy(), atol=1e-6)


# ---- bench.py's N > 1 entry point: it must start by itself (the reference spawns its own ranks: train.py:52-59,315) ----------
def _run_bench(cmd, env_extra=None, timeout=240):
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable] + cmd, cwd=repo, env=env, capture_output=True, text=True, timeout=timeout)
    return p


@pytest.mark.timeout(300)
def test_bench_gpus2_spawns_its_own_ranks_and_joins_the_process_group():
    """``python bench.py --gpus 2`` with NO launcher: bench.py spawns two ranks, they rendezvous on 127.0.0.1 (gloo here,
    RCCL on a GPU node) and one all-reduce goes round; rank 0 prints one JSON line.  Round 2's bench.py aborted here."""
    import json
    p = _run_bench(["bench.py", "--gpus", "2", "--backend", "gloo", "--rendezvous-only"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec == {"rendezvous": "ok", "world": 2, "backend": "gloo", "allreduce": 3.0, "launcher": "self-spawned"}


@pytest.mark.timeout(300)
def test_bench_gpus2_under_torch_distributed_run():
    """The driver's form: ``python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`` keeps working."""
    import json
    p = _run_bench(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--backend", "gloo", "--rendezvous-only"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["world"] == 2 and rec["allreduce"] == 3.0 and rec["launcher"] == "torch.distributed.run"


def test_bench_refuses_a_launcher_that_disagrees_with_gpus_flag():
    p = _run_bench(["bench.py", "--gpus", "2", "--backend", "gloo", "--rendezvous-only"], env_extra={"WORLD_SIZE": "1", "RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)


# ---- slab exchange of the Vis source-view shard: reduce-scatter of 16-bit shares + neighbour halos ----------------------------
def _slab_worker(rank, world, port, shape, axis, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shares = [torch.randint(-8, 9, shape, generator=torch.Generator().manual_seed(100 + r)).to(torch.float16) for r in range(world)]
        total = torch.stack(shares).sum(0)                      # small integers: exact in fp16 in any summation order
        ext, lo, a, b = pdist.reduce_to_slab(shares[rank], axis, halo=pdist.FUSE_HALO)
        E = shape[axis]
        S = pdist.slab_size(E, world)
        ok = True
        if ext is None:
            ok = a == b == rank * S and rank * S >= E
            rows = torch.zeros((shape[0], 1, 0, shape[3]), dtype=torch.float32)
        else:
            hi = lo + ext.shape[axis]
            ok = (a == rank * S and b == min(E, a + S) and lo == max(0, a - pdist.FUSE_HALO) and hi == min(E, b + pdist.FUSE_HALO)
                  and torch.equal(ext, total.narrow(axis, lo, hi - lo)) and ext.is_contiguous())
            rows = total.float().narrow(axis, a, b - a).sum(dim=(1 if axis == 2 else 2, 4)).unsqueeze(1) if axis == 2 else None
        gathered = None
        if axis == 2:
            full = pdist.gather_rows(rows, E, S)
            gathered = bool(torch.equal(full, total.float().sum(dim=(1, 4)).unsqueeze(1)))
        q.put((rank, bool(ok), gathered, (lo, a, b)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world,shape,axis", [(2, (1, 32, 6, 5, 8), 1), (3, (2, 4, 50, 3, 8), 2), (4, (1, 6, 50, 3, 8), 2),
                                              (5, (1, 2, 32, 3, 8), 2)])     # 32 rows / 5 ranks: slabs of 8, the tail rank owns none
def test_reduce_to_slab_sums_and_exchanges_halos(world, shape, axis):
    """Every rank ends up with the SUM of all shares on its owned units + FUSE_HALO units per inner side, contiguous in the
    original layout; uneven extents (50 rows over 3 / 4 ranks: the tail rank owns fewer rows, or none) included; gather_rows
    reassembles per-row results on every rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_slab_worker, args=(r, world, port, shape, axis, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
    if axis == 2:
        assert all(r[2] for r in res), res
    spans = [r[3] for r in res]
    assert spans[0][1] == 0 and max(s[2] for s in spans) == shape[axis]


def test_slab_axis_prefers_the_thicker_slab_and_refuses_thin_ones():
    assert pdist.slab_axis(256, 144, 8) == (1, 32)          # configuration 5, stage 1: depth slabs of 32 planes
    assert pdist.slab_axis(32, 288, 8) == (2, 36)           # stage 2: row slabs
    assert pdist.slab_axis(16, 576, 8) == (2, 72)           # stage 3
    assert pdist.slab_axis(16, 8, 2) == (1, 8)
    assert pdist.slab_axis(8, 12, 2) is None                # tiny fixture: replicated path
    assert pdist.slab_size(50, 4) == 14 and pdist.slab_size(50, 3) == 18


def test_bench_sharded_legs_run_under_a_wall_clock_budget():
    """bench.py's N > 1 sharded legs are the one part of the run that moves data between GPUs; a stuck collective there must not
    cost the headline line: the legs run in a helper thread and the caller gets an error record after the budget."""
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    dev = torch.device("cpu")
    ok, done = bench.sharded_legs_bounded(None, dev, 2, 0, 5.0, legs=lambda d, de, w, r: {"depth": {"speedup_vs_1gpu": 1.5}})
    assert done and ok == {"depth": {"speedup_vs_1gpu": 1.5}}
    t0 = time.time()
    res, done = bench.sharded_legs_bounded(None, dev, 2, 1, 0.3, legs=lambda d, de, w, r: time.sleep(30))
    assert not done and "did not finish" in res["error"] and "rank 1" in res["error"] and time.time() - t0 < 5
    res, done = bench.sharded_legs_bounded(None, dev, 2, 0, 5.0, legs=lambda d, de, w, r: 1 / 0)
    assert done and res["error"].startswith("ZeroDivisionError")
