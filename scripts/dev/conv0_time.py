#!/usr/bin/env python3
"""Dev: MVSNet conv0 (32 -> 8, 192 x 128 x 160, fp16) on the sweep kernel: median of 7 x 20 launches; a checksum of the output.
python scripts/dev/conv0_time.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wild_deep_mvs_amd import _lib as L, ops  # noqa: E402  (PSCV_LIB selects an A/B build)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
g = torch.Generator().manual_seed(0)
x = (torch.randn(B, 192, 128, 160, 32, generator=g) * 0.5).to(torch.float16).cuda()
w = torch.randn(8, 32, 3, 3, 3, generator=g) / (27 * 32) ** 0.5
layer = ops.Conv3dLayer.build(w, kind=L.CONV_S1, device="cuda", relu=True, dtype=torch.float16)
out = torch.empty(B, 192, 128, 160, 8, dtype=torch.float16, device="cuda")
ts = []
for rep in range(7):
    for _ in range(2):
        ops.conv3d(x, layer, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv3d(x, layer, out=out)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20 * 1e3)
ts.sort()
print(f"conv0 B={B}: median {ts[3]:.1f} us  min {ts[0]:.1f}  max {ts[-1]:.1f}   checksum {out.float().double().sum().item():.6f} {int(out.view(torch.int16).long().sum())}")
