"""Dev probe: do the warp+cost kernel (vector-ALU bound) and the 32->8 sweep conv (HBM / LDS / MFMA) overlap when they run on
two streams?  Times warp alone, conv0 alone, both back to back on one stream, and both concurrently on two streams."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from wild_deep_mvs_amd import _lib as L, ops

dev = torch.device("cuda", 0)
net, sd, feats, feats_cl, proj_d, dv_d, proj, dv = bench.build_inputs(dev, 0, torch.float16)
cams = ops.proj_cams_device(proj_d.to(torch.float32).contiguous(), 0)
cost_a = ops.warp_cost(feats_cl[0], feats_cl[1:], cams, dv_d, out_dtype=torch.float16)
cost_b = cost_a.clone()
l0 = net.cost_regularization.engine_layers(torch.float16)["conv0"]
out0 = torch.empty(1, bench.D, bench.h, bench.w, 8, dtype=torch.float16, device=dev)
s2 = torch.cuda.Stream()

def warp(): ops.warp_cost(feats_cl[0], feats_cl[1:], cams, dv_d, out_dtype=torch.float16, out=cost_a)
def conv(): ops.conv3d(cost_b, l0, out=out0)
def serial(): warp(); conv()
def concurrent():
    s2.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s2):
        conv()
    warp()
    torch.cuda.current_stream().wait_stream(s2)

def timed(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for name, fn in (("warp", warp), ("conv0", conv), ("serial", serial), ("two streams", concurrent)):
    print(f"{name:12s} {timed(fn):8.1f} us")
