#!/usr/bin/env python3
"""Dev: which side of the two-rank MVSNet depth-plane shard test is not reproducible?  Two ranks on cuda:0 over gloo (the set-up of
tests/test_gpu_dist.py); each iteration runs the unsharded and the sharded forward and compares each with ITS OWN first result bit
for bit, and the two with each other; bad pixels are reported with their bounding box.  A third process can keep the GPU busy
(--load) the way the pytest parent's earlier tests do.
Usage: python scripts/dev/depth_shard_race.py [--iters 30] [--load]"""
import argparse
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def worker(rank, world, port, iters, q, skip_sharded=False, sync=False, cases=(0, 1, 2)):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wild_deep_mvs_amd import synthetic
    from wild_deep_mvs_amd.models.MVSNet.model import MVSNet
    dev = torch.device("cuda", 0)
    lines = []
    all_cases = [("variance", 1, 3, 128, 160, 48), ("variance", 2, 3, 128, 160, 48), ("variance", 1, 5, 512, 640, 192)]
    for (agg, B, V, H, W, D) in [all_cases[i] for i in cases]:
        net = MVSNet(agg)
        net.load_state_dict(synthetic.sharpened_state_dict("mvsnet", synthetic.template_of(net), seed=0))
        net = net.to(dev).eval()
        net.num_depth = D
        net.graph_replay = False
        scene = {k: v.to(dev) for k, v in synthetic.make_scene(B, V, H, W, seed=7).items()}
        call = lambda: net(scene["imgs"], scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"])["depth"].clone()
        first = {}
        for it in range(iters):
            with torch.no_grad():
                net.set_depth_group(None)
                u = call()
                if sync:
                    torch.cuda.synchronize()
                net.set_depth_group(None if skip_sharded else dist.group.WORLD)
                s = call()
                if sync:
                    torch.cuda.synchronize()
            for name, x in (("unsharded", u), ("sharded", s)):
                if name not in first:
                    first[name] = x
                elif not torch.equal(first[name], x):
                    bad = (first[name] - x).abs() > 0
                    idx = bad.nonzero()
                    lines.append(f"rank {rank} case {B}x{V}x{H}x{W} D={D} it {it}: {name} differs from its first run on {int(bad.sum())} px, "
                                 f"max {float((first[name] - x).abs().max()):.3e}, box b {idx[:,0].min().item()}-{idx[:,0].max().item()} "
                                 f"y {idx[:,1].min().item()}-{idx[:,1].max().item()} x {idx[:,2].min().item()}-{idx[:,2].max().item()}")
            rel = float((u - s).abs().max() / u.abs().max())
            if rel > 2e-6:
                lines.append(f"rank {rank} case {B}x{V}x{H}x{W} D={D} it {it}: sharded vs unsharded max rel {rel:.3e}")
        lines.append(f"rank {rank} case {B}x{V}x{H}x{W} D={D}: {iters} iterations done")
    q.put((rank, lines))
    dist.destroy_process_group()


def load(stop):
    x = torch.randn(4096, 4096, device="cuda")
    while not stop.is_set():
        for _ in range(20):
            x = (x @ x).tanh()
        torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--load", action="store_true")
    ap.add_argument("--skip-sharded", action="store_true", help="both calls of an iteration unsharded (gloo initialised but unused)")
    ap.add_argument("--sync", action="store_true", help="device synchronisation after every forward")
    ap.add_argument("--cases", type=int, nargs="*", default=[0, 1, 2])
    args = ap.parse_args()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    stop = ctx.Event()
    lp = ctx.Process(target=load, args=(stop,)) if args.load else None
    if lp:
        lp.start()
    procs = [ctx.Process(target=worker, args=(r, 2, port, args.iters, q, args.skip_sharded, args.sync, tuple(args.cases))) for r in range(2)]
    for p in procs:
        p.start()
    for _ in range(2):
        rank, lines = q.get(timeout=900)
        print("\n".join(lines), flush=True)
    for p in procs:
        p.join(60)
    if lp:
        stop.set(); lp.join(30)


if __name__ == "__main__":
    main()
