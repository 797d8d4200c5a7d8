import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wild_deep_mvs_amd import synthetic
from wild_deep_mvs_amd.models.CVP_MVSNet.frontend import Frontend
net = Frontend(); net.load_state_dict(synthetic.sharpened_state_dict("cvp", synthetic.template_of(net), seed=0)); net = net.cuda().eval()
sc = synthetic.make_scene(1, 3, 128, 160, seed=4); sc["t"] = sc["t"] * 8
a = [sc[k].cuda() for k in ("imgs", "K", "R", "t", "depth_min", "depth_max")]
with torch.no_grad():
    for _ in range(2): net(*a, nscale=2)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            out = net(*a, nscale=2)
        print("captured ok")
    except Exception:
        traceback.print_exc()
