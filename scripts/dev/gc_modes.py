#!/usr/bin/env python3
"""Dev: staging-mode histogram of the LDS-staged group-correlation kernel over one forward of a Vis configuration with warp_gc_lds = 2."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import run_configs as RC
from wild_deep_mvs_amd import _lib as L, ops, synthetic
cid = int(sys.argv[1]) if len(sys.argv) > 1 else 5
cfg = RC.CONFIGS[cid]
net = RC.build(cfg["arch"]); cfg["setup"](net)
scene = synthetic.make_scene(1, cfg["V"], cfg["H"], cfg["W"], seed=cid)
dev = {k: v.cuda() for k, v in scene.items()}
L.set_tuning("warp_gc_lds", 2)
orig = ops.warp_cost
fn = L.lib().pscv_debug_wl_mode_hist
fn.argtypes, fn.restype = [ctypes.c_void_p], None
def wrapped(ref, srcs, cams, dv, **kw):
    hist = torch.zeros(16, dtype=torch.int32, device="cuda")
    fn(hist.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = orig(ref, srcs, cams, dv, **kw); e1.record(); torch.cuda.synchronize(); fn(None)
    hm = hist.view(4, 4).sum(0).tolist()
    tot = max(1, sum(hm))
    print(f"warp_cost D={dv.shape[1]} planes{'/pixel' if dv.dim() == 4 else ''} {tuple(ref.shape[1:3])} x {len(srcs)} views: {e0.elapsed_time(e1) * 1e3:.0f} us; "
          f"(block, view) modes DIRECT {hm[0] / tot:.2f} GEN {hm[1] / tot:.2f} FAST {hm[2] / tot:.2f} ZERO {hm[3] / tot:.2f}", flush=True)
    return out
with torch.no_grad():
    net.graph_replay = False
    net(dev["imgs"], dev["K"], dev["R"], dev["t"], dev["depth_min"], dev["depth_max"], **cfg["kw"])
    import wild_deep_mvs_amd.models.VisMVSNet.model_cas as MC
    MC.ops.warp_cost = wrapped
    for gc in (2, 0):
        L.set_tuning("warp_gc_lds", gc)
        print("warp_gc_lds =", gc)
        net(dev["imgs"], dev["K"], dev["R"], dev["t"], dev["depth_min"], dev["depth_max"], **cfg["kw"])
