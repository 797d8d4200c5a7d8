"""Parity-pair deconv (16 -> 8 transposed, stride 2, + skip) at Vis / MVSNet sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wild_deep_mvs_amd import _lib as L
if os.environ.get("PSCV_LIB"):
    L.LIB_PATH = os.environ["PSCV_LIB"]
from wild_deep_mvs_amd import ops
g = torch.Generator().manual_seed(0)
for (D, H, W) in [(96, 64, 80), (128, 72, 100), (8, 288, 400), (16, 128, 160)]:
    w = torch.randn(16, 8, 3, 3, 3, generator=g) / 20
    layer = ops.Conv3dLayer.build(w, kind=L.CONV_T2, transposed=True, device="cuda", relu=True, dtype=torch.float16)
    x = (torch.randn(1, D, H, W, 16, generator=g) * 0.5).to(torch.float16).cuda()
    skip = (torch.randn(1, 2 * D, 2 * H, 2 * W, 8, generator=g) * 0.5).to(torch.float16).cuda()
    for _ in range(3):
        y = ops.conv3d(x, layer, skip=skip)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        y = ops.conv3d(x, layer, skip=skip)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    by = x.numel() * 2 + 2 * y.numel() * 2
    print(f"{D}x{H}x{W} -> x2: {us:8.1f} us  {by / us / 1e3:7.0f} GB/s   checksum {float(y.float().abs().sum()):.6e}")
