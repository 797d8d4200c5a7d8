"""Shapes + per-call event times of the conv3d launches of one configuration (eager)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import run_configs as RC
from wild_deep_mvs_amd import ops, synthetic
cid = int(sys.argv[1]); cmin = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg = RC.CONFIGS[cid]
net = RC.build(cfg["arch"]); cfg["setup"](net); net.graph_replay = False
scene = synthetic.make_scene(1, cfg["V"], cfg["H"], cfg["W"], seed=cid)
dev = {k: v.cuda() for k, v in scene.items()}
call = lambda: net(dev["imgs"], dev["K"], dev["R"], dev["t"], dev["depth_min"], dev["depth_max"], **cfg["kw"])
with torch.no_grad():
    call(); call(); torch.cuda.synchronize()
    orig = ops.conv3d
    rec = []
    def spy(x, layer, **kw):
        if layer.c_in >= cmin:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); y = orig(x, layer, **kw); e1.record()
            rec.append((tuple(x.shape), layer.c_in, layer.c_out, layer.kind, e0, e1))
            return y
        return orig(x, layer, **kw)
    ops.conv3d = spy
    call(); torch.cuda.synchronize()
for shp, ci, co, k, e0, e1 in rec:
    us = e0.elapsed_time(e1) * 1e3
    vox = shp[0] * shp[1] * shp[2] * shp[3]
    tf = 2.0 * 27 * ci * co * vox / (us * 1e-6) / 1e12 if k in (0, 3) else 0
    print(f"{shp} {ci}->{co} kind {k}: {us:7.1f} us  {tf:6.1f} TF")
