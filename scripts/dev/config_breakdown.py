"""Per-kernel time table of one eager forward of a BASELINE configuration (HIP events per launch + torch-side remainder)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import run_configs as RC
from wild_deep_mvs_amd import ops, synthetic
for cid in [int(x) for x in sys.argv[1:]] or [4, 5]:
    cfg = RC.CONFIGS[cid]
    net = RC.build(cfg["arch"]); cfg["setup"](net); net.graph_replay = False
    scene = synthetic.make_scene(1, cfg["V"], cfg["H"], cfg["W"], seed=cid)
    if "bscale" in cfg:
        scene["t"] = scene["t"] * cfg["bscale"]
    dev = {k: v.cuda() for k, v in scene.items()}
    call = lambda: net(dev["imgs"], dev["K"], dev["R"], dev["t"], dev["depth_min"], dev["depth_max"], **cfg["kw"])
    with torch.no_grad():
        call(); call(); torch.cuda.synchronize()
        t0 = time.perf_counter(); call(); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
        with ops.EventTimer() as tm:
            call()
        det = tm.detail()
    tot = sum(v["ms"] for v in det.values())
    print(f"== config {cid}: wall {wall:.2f} ms, engine kernels {tot:.2f} ms in {sum(v['launches'] for v in det.values())} launches")
    for k, v in sorted(det.items(), key=lambda kv: -kv[1]["ms"]):
        gbs = v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["bytes"] else 0
        tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["flops"] else 0
        print(f"  {k:28s} x{v['launches']:3d}  {v['ms']:7.3f} ms  {100 * v['ms'] / tot:5.1f}%  avg {v['ms'] / v['launches'] * 1e3:7.1f} us  {gbs:7.0f} GB/s  {tf:6.1f} TF")
    del net, dev
    torch.cuda.empty_cache()
