"""What do the channel-slice accesses of the Vis U-Net's cat buffer cost?  8->8 sweep / deconv writing dense vs into [.., 16] slices,
stride-2 conv reading dense vs a slice; 256 x 144 x 200 (configuration 5, stage 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wild_deep_mvs_amd import _lib as L, ops
g = torch.Generator().manual_seed(0)
D, H, W = 256, 144, 200
dt = torch.float16
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
x8 = (torch.randn(1, D, H, W, 8, generator=g) * 0.5).to(dt).cuda()
cat = torch.zeros(1, D, H, W, 16, dtype=dt, device="cuda")
cat[..., 8:] = x8
w88 = torch.randn(8, 8, 3, 3, 3, generator=g) / 15
l88 = ops.Conv3dLayer.build(w88, kind=L.CONV_S1, device="cuda", relu=True, dtype=dt)
print(f"8->8 sweep, dense out      : {timeit(lambda: ops.conv3d(x8, l88)):7.1f} us")
print(f"8->8 sweep, out = cat[8:16]: {timeit(lambda: ops.conv3d(x8, l88, out=cat, out_coff=8)):7.1f} us")
print(f"8->8 sweep, in = cat[8:16] : {timeit(lambda: ops.conv3d(cat, l88, in_coff=8)):7.1f} us")
w832 = torch.randn(32, 8, 3, 3, 3, generator=g) / 15
l832 = ops.Conv3dLayer.build(w832, kind=L.CONV_S2, device="cuda", relu=True, dtype=dt)
print(f"8->32 s2, dense in         : {timeit(lambda: ops.conv3d(x8, l832)):7.1f} us")
print(f"8->32 s2, in = cat[8:16]   : {timeit(lambda: ops.conv3d(cat, l832, in_coff=8)):7.1f} us")
x16h = (torch.randn(1, D // 2, H // 2, W // 2, 16, generator=g) * 0.5).to(dt).cuda()
wd = torch.randn(16, 8, 3, 3, 3, generator=g) / 20
ld = ops.Conv3dLayer.build(wd, kind=L.CONV_T2, transposed=True, device="cuda", dtype=dt)
print(f"deconv 16->8, dense out    : {timeit(lambda: ops.conv3d(x16h, ld)):7.1f} us")
print(f"deconv 16->8, out=cat[0:8] : {timeit(lambda: ops.conv3d(x16h, ld, out=cat, out_coff=0)):7.1f} us")
w168 = torch.randn(8, 16, 3, 3, 3, generator=g) / 20
l168 = ops.Conv3dLayer.build(w168, kind=L.CONV_S1, device="cuda", dtype=dt)
print(f"16->8 sweep (post), cat in : {timeit(lambda: ops.conv3d(cat, l168)):7.1f} us")
