"""Dev aid: Vis-MVSNet forward time with the 2-D extractor on the engine vs PyTorch-ROCm."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wild_deep_mvs_amd import ops, synthetic
from wild_deep_mvs_amd.models.VisMVSNet.frontend import Frontend

def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

net = Frontend(); net.load_state_dict(synthetic.sharpened_state_dict("vis", synthetic.template_of(net), seed=0)); net = net.cuda().eval()
for (V, H, W, kw) in ((5, 512, 640, dict(depth_nums=[192, 32, 16], interval_scales=[128 / 192, 1, 0.5])),
                      (5, 512, 640, dict(depth_nums=[64, 32, 16], interval_scales=[2, 1, 0.5])),
                      (9, 1152, 1600, dict(depth_nums=[256, 32, 16], interval_scales=[0.5, 1, 0.5]))):
    sc = {k: v.cuda() for k, v in synthetic.make_scene(1, V, H, W, seed=0).items()}
    x = torch.cat(list(torch.unbind(sc["imgs"], 1)), 0)
    with torch.no_grad():
        for eng in ("torch", "pscv"):
            net.feature_engine = eng
            t_all = timeit(lambda: net(sc["imgs"], sc["K"], sc["R"], sc["t"], sc["depth_min"], sc["depth_max"], **kw))
            t_fe = timeit(lambda: (net.model.feat_ext.forward_engine(x, torch.float16) if eng == "pscv" else net.model.feat_ext(x)))
            print(f"V={V} {H}x{W} D={kw['depth_nums']}: feature_engine={eng:5s} forward {t_all:7.2f} ms, FeatExt alone {t_fe:6.2f} ms")

from wild_deep_mvs_amd.graph import GraphedModel
net.feature_engine = "pscv"
g = GraphedModel(net)
for (V, H, W, kw) in ((5, 512, 640, dict(depth_nums=[64, 32, 16], interval_scales=[2, 1, 0.5])),
                      (9, 1152, 1600, dict(depth_nums=[256, 32, 16], interval_scales=[0.5, 1, 0.5]))):
    sc = {k: v.cuda() for k, v in synthetic.make_scene(1, V, H, W, seed=0).items()}
    with torch.no_grad():
        t_e = timeit(lambda: net(sc["imgs"], sc["K"], sc["R"], sc["t"], sc["depth_min"], sc["depth_max"], **kw))
        t_g = timeit(lambda: g(sc["imgs"], sc["K"], sc["R"], sc["t"], sc["depth_min"], sc["depth_max"], **kw))
    print(f"V={V} {H}x{W} D={kw['depth_nums']}: eager {t_e:7.2f} ms, hipGraph replay {t_g:7.2f} ms")
