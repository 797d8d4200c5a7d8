import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wild_deep_mvs_amd import _lib as L, ops
dt = torch.float16; dev = "cuda"
D, h, w = 192, 128, 160
g = torch.Generator().manual_seed(0)
x = (torch.randn(1, D, h, w, 8, generator=g) * 0.5).to(dt).to(dev)
wt = torch.randn(1, 8, 3, 3, 3, generator=g) / (27 * 8) ** 0.5
layer = ops.Conv3dLayer.build(wt, kind=L.CONV_S1, device=dev, relu=False, dtype=dt)
out = torch.empty(1, D, h, w, 1, dtype=torch.float32, device=dev)

def gtime(fn, n=20, reps=10):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s, capture_error_mode="thread_local"):
            for _ in range(n): fn()
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps): gr.replay()
        e1.record(s)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3

for nb in (2, 1):
    L.set_tuning("c1_nb", nb)
    print(f"nb={nb}: {gtime(lambda: ops.conv3d(x, layer, out=out)):7.1f} us")
