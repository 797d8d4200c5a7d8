#!/usr/bin/env python3
"""Does any kernel of a forward read memory it did not write?  Run the forward with the caching allocator's free blocks filled with
zeros, then with NaN patterns / large finite values, and compare outputs (they must be identical)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from run_configs import CONFIGS, build
from wild_deep_mvs_amd import synthetic
cid = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = CONFIGS[cid]
net = build(cfg["arch"]); cfg["setup"](net)
scene = synthetic.make_scene(1, cfg["V"], cfg["H"], cfg["W"], seed=cid)
if "bscale" in cfg:
    scene["t"] = scene["t"] * cfg["bscale"]
dev = {k: v.cuda() for k, v in scene.items()}
call = lambda: net(dev["imgs"], dev["K"], dev["R"], dev["t"], dev["depth_min"], dev["depth_max"], **cfg["kw"])

def pollute(kind):
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    blocks = []
    # fill what the allocator has cached plus a few GiB of fresh memory with the pattern, then hand it all back to the cache
    for sz in (1 << 30,) * 6 + (1 << 26,) * 16 + (1 << 20,) * 64:
        t = torch.empty(sz // 4, dtype=torch.float32, device="cuda")
        if kind == "zero": t.zero_()
        elif kind == "nan": t.fill_(float("nan"))
        elif kind == "big": t.fill_(3.0e38)
        elif kind == "ones": t.view(torch.int32).fill_(-1)
        blocks.append(t)
    torch.cuda.synchronize()
    del blocks

outs = {}
with torch.no_grad():
    call(); call()
    for kind in ("zero", "nan", "big", "ones", "zero"):
        pollute(kind)
        o = call()
        torch.cuda.synchronize()
        outs.setdefault(kind, []).append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items() if k in ("depth", "photometric_confidence")})
ref = outs["zero"][0]
for kind, lst in outs.items():
    for j, o in enumerate(lst):
        for k in ref:
            df = (o[k].float() - ref[k].float()).abs()
            nan = int(torch.isnan(o[k]).sum())
            print(f"config {cid} free-memory pattern {kind}[{j}] {k}: max diff {float(torch.nan_to_num(df).max()):.4g}  differing {float((df > 0).float().mean()):.3g}  nan {nan}")
