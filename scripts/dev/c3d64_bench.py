"""conv3d 64 -> 64 (CVP's widest 3-D layer) at config 4's sizes: event times per variant; run under rocprofv3 --pmc for counters."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wild_deep_mvs_amd import _lib as L, ops
shape = tuple(int(v) for v in sys.argv[1].split("x")) if len(sys.argv) > 1 else (4, 512, 640)
cout = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
D, H, W = shape
g = torch.Generator().manual_seed(0)
x = torch.randn(1, D, H, W, 64, generator=g).to(torch.float16).cuda()
w = torch.randn(cout, 64, 3, 3, 3, generator=g) / (27 * 64) ** 0.5
layer = ops.Conv3dLayer.build(w, kind=L.CONV_S1, device="cuda", dtype=torch.float16, relu=True)
for tall in (0, 1):
    L.set_tuning("conv_tall64", tall)
    for _ in range(3):
        y = ops.conv3d(x, layer)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        y = ops.conv3d(x, layer)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    tf = 2.0 * 27 * 64 * cout * D * H * W / (us * 1e-6) / 1e12
    print(f"64->{cout} {shape} tall={tall}: {us:8.1f} us  {tf:6.1f} TF  ({tf / 2500:.2f} of the dense f16 MFMA peak)")
