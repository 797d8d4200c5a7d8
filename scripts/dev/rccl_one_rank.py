#!/usr/bin/env python3
"""RCCL with a world of one rank on a one-GPU box: which of the calls the sharded models issue work there?  Prints as it goes
(run under `timeout`)."""
import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

def say(*a):
    print(f"[{time.time() - T0:6.1f}s]", *a, flush=True)

T0 = time.time()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
say("init_process_group nccl ...")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
say("ok; backend", dist.get_backend())
x = torch.ones(1024, device=dev)
dist.all_reduce(x); torch.cuda.synchronize(); say("all_reduce fp32 ok")
h = torch.ones(1024, device=dev, dtype=torch.float16)
dist.all_reduce(h); torch.cuda.synchronize(); say("all_reduce fp16 ok")
b = torch.ones(1024, device=dev, dtype=torch.bfloat16)
dist.all_reduce(b); torch.cuda.synchronize(); say("all_reduce bf16 ok")
dist.all_reduce(x, op=dist.ReduceOp.MAX); torch.cuda.synchronize(); say("all_reduce MAX ok")
d = torch.ones(4, device=dev, dtype=torch.float64)
dist.all_reduce(d, op=dist.ReduceOp.MAX); torch.cuda.synchronize(); say("all_reduce f64 MAX ok")
bufs = [torch.empty_like(x)]
dist.all_gather(bufs, x); torch.cuda.synchronize(); say("all_gather ok")
own = torch.empty(1024, device=dev, dtype=torch.float16)
dist.reduce_scatter_tensor(own, h); torch.cuda.synchronize(); say("reduce_scatter_tensor ok")
dist.barrier(); torch.cuda.synchronize(); say("barrier ok")
from wild_deep_mvs_amd import dist as pd
ext, lo, a, bb = pd.reduce_to_slab(torch.arange(24, dtype=torch.float32, device=dev).reshape(1, 6, 4).half(), 1, None, halo=2)
say("reduce_to_slab ok", lo, a, bb)
import bench
for mode in (sys.argv[1:] or ["mvsnet_depth", "depth", "view"]):
    say("sharded leg", mode, "...")
    res = bench.sharded_legs(dist, dev, 1, 0, reps=1, only=(mode,))
    r = res.get(mode, {})
    say("   ->", {k: r.get(k) for k in ("error", "depth_rel_l1_vs_unsharded", "ms_per_forward_1gpu", "ms_per_forward_sharded")}, [(c["collective"], c["bytes_per_rank"], c["calls"]) for c in r.get("collectives", [])])
dist.destroy_process_group()
say("done")
