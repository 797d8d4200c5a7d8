"""Dev aid: wall-clock of a few host-side pieces on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wild_deep_mvs_amd import ops, synthetic

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

from wild_deep_mvs_amd.models.MVSNet.model import MVSNet
net = MVSNet("variance"); net.load_state_dict(synthetic.sharpened_state_dict("mvsnet", synthetic.template_of(net), seed=0)); net = net.cuda().eval()
sc = {k: v.cuda() for k, v in synthetic.make_scene(1, 5, 512, 640, seed=0).items()}
with torch.no_grad():
    print("mvsnet forward eager ms", timeit(lambda: net(sc["imgs"], sc["K"], sc["R"], sc["t"], sc["depth_min"], sc["depth_max"])))
    imgs = list(torch.unbind(sc["imgs"], 1))
    print("  features (pscv) ms", timeit(lambda: net.extract_features_cl(imgs)))
    with ops.EventTimer() as tm:
        for _ in range(10): net(sc["imgs"], sc["K"], sc["R"], sc["t"], sc["depth_min"], sc["depth_max"])
    s = tm.summary(); print("  kernel sum ms/forward", sum(v[1] for v in s.values()) / 10, "launches", sum(v[0] for v in s.values()) / 10)

from wild_deep_mvs_amd.models.CVP_MVSNet.models.modules import calDepthHypo
for H, W in ((1024, 1280), (512, 640), (128, 160)):
    B = 1
    scene = synthetic.make_scene(B, 3, H, W, seed=4); scene["t"] = scene["t"] * 8
    row = torch.tensor([0., 0., 0., 1.])
    ref_ex = torch.cat((torch.cat((scene["R"][:, 0], scene["t"][:, 0]), 2), row.view(1, 1, 4).expand(B, 1, 4)), 1).cuda()
    src_ex = torch.cat((torch.cat((scene["R"][:, 1:], scene["t"][:, 1:]), 3), row.view(1, 1, 1, 4).expand(B, 2, 1, 4)), 2).cuda()
    K = scene["K"].cuda(); depth = (2.5 + 3 * torch.rand(B, H, W)).cuda()
    dmin, dmax = scene["depth_min"][:, 0].cuda(), scene["depth_max"][:, 0].cuda()
    print(f"calDepthHypo {H}x{W} ms", timeit(lambda: calDepthHypo(depth, K[:, 0], K[:, 1:], ref_ex, src_ex, dmin, dmax, 0)))
    with ops.EventTimer() as tm:
        calDepthHypo(depth, K[:, 0], K[:, 1:], ref_ex, src_ex, dmin, dmax, 0)
    print("   kernel ms", {k: round(v[1], 3) for k, v in tm.summary().items()})
