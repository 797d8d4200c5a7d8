"""Dev: GPU-side kernel time of a steady-state Vis / CVP / MVSNet training step (torch.profiler over 3 steps after warm-up)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wild_deep_mvs_amd import synthetic
arch = sys.argv[1] if len(sys.argv) > 1 else "vis"
H, W, V = 512, 640, 5
if arch == "vis":
    from wild_deep_mvs_amd.models.VisMVSNet.frontend import Frontend
    net = Frontend(); kw = dict(depth_nums=[64, 32, 16], interval_scales=[2.0, 1.0, 0.5]); down = 2
    net.depth_nums, net.interval_scales = kw["depth_nums"], kw["interval_scales"]
    key = "vis"
else:
    from wild_deep_mvs_amd.models.MVSNet.model import MVSNet
    net = MVSNet("variance"); kw = {}; down = 4; key = "mvsnet"
net.load_state_dict(synthetic.train_state_dict(key, synthetic.template_of(net), seed=0))
net = net.cuda().train()
if os.environ.get("PSCV_FE"):
    net.feature_engine_train = os.environ["PSCV_FE"]      # MVSNet: 2-D extractor on the engine (training.FeatureNetFn)
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
scene = synthetic.make_scene(1, V, H, W, seed=0)
gt, mask = synthetic.train_target(scene, H // down, W // down)
dev = {k: v.cuda() for k, v in scene.items() if isinstance(v, torch.Tensor)}
gt, mask = gt.cuda(), mask.cuda()
def step():
    opt.zero_grad(set_to_none=True)
    out = net(dev["imgs"], dev["K"], dev["R"], dev["t"], dev["depth_min"], dev["depth_max"], **kw)
    loss = (synthetic.vis_supervised_loss(out, gt, mask, dev["depth_min"], dev["depth_max"], V) if arch == "vis"
            else synthetic.supervised_loss(out["depth"], gt, mask, dev["depth_min"], dev["depth_max"]))
    loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
ev = [e for e in prof.key_averages() if e.device_time_total > 0]
tot = sum(e.device_time_total for e in ev) / 3e3
print(f"{arch}: GPU kernel time per step {tot:.2f} ms")
kern = [e for e in ev if not e.key.startswith(("aten::", "autograd::")) and "Backward" not in e.key and "Fn" not in e.key[-4:]]
print(f"  kernels only: {sum(e.device_time_total for e in kern) / 3e3:.2f} ms in {sum(e.count for e in kern) // 3} launches per step")
for e in sorted(kern, key=lambda e: -e.device_time_total)[:22]:
    print(f"  x{e.count // 3:5d}  {e.device_time_total / 3e3:8.3f} ms  {e.key[:120]}")
cpu = sorted(prof.key_averages(), key=lambda e: -e.count)[:14]
for e in cpu:
    print(f"  cpu x{e.count // 3:5d} {e.self_cpu_time_total / 3e3:8.3f} ms  {e.key[:80]}")
