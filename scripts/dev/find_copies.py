"""Which Python lines of an eager forward issue device-to-device memcpys / tiny ATen kernels (torch.profiler, with stacks)."""
import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
import run_configs as RC
from wild_deep_mvs_amd import synthetic
cid = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cfg = RC.CONFIGS[cid]
net = RC.build(cfg["arch"]); cfg["setup"](net); net.graph_replay = False
scene = synthetic.make_scene(1, cfg["V"], cfg["H"], cfg["W"], seed=cid)
dev = {k: v.cuda() for k, v in scene.items()}
call = lambda: net(dev["imgs"], dev["K"], dev["R"], dev["t"], dev["depth_min"], dev["depth_max"], **cfg["kw"])
with torch.no_grad():
    call(); call(); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
        call(); torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::"):
        continue
    if ev.name in ("aten::copy_", "aten::fill_", "aten::mul", "aten::add", "aten::sub", "aten::div", "aten::cat", "aten::clone",
                   "aten::contiguous", "aten::_to_copy", "aten::zeros", "aten::arange", "aten::clamp", "aten::exp", "aten::stack"):
        st = [s for s in ev.stack if "wild_deep_mvs_amd" in s or "run_configs" in s]
        cnt[(ev.name, st[0] if st else "?")] += 1
for (name, where), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{n:4d}  {name:18s} {where}")
