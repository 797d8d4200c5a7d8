#!/usr/bin/env python3
"""Dev: ten launches of the 64 -> 64 stride-1 3-D layer at CVP's finest refinement size (4 x 512 x 640) for a rocprofv3 --pmc run;
`conv_wide` = argv[1] (0 brick, 1 wide, 3 reduction-split variant)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wild_deep_mvs_amd import _lib as L  # noqa: E402
if os.environ.get("PSCV_LIB"):
    L.LIB_PATH = os.environ["PSCV_LIB"]
from wild_deep_mvs_amd import ops  # noqa: E402

L.set_tuning("conv_wide", int(sys.argv[1]) if len(sys.argv) > 1 else 1)
g = torch.Generator().manual_seed(0)
x = (torch.randn(1, 4, 512, 640, 64, generator=g) * 0.5).to(torch.float16).cuda()
w = torch.randn(64, 64, 3, 3, 3, generator=g) / (27 * 64) ** 0.5
layer = ops.Conv3dLayer.build(w, kind=L.CONV_S1, device="cuda", relu=True, dtype=torch.float16)
out = torch.empty(1, 4, 512, 640, 64, dtype=torch.float16, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    ops.conv3d(x, layer, out=out)
e0.record()
for _ in range(10):
    ops.conv3d(x, layer, out=out)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 10 * 1e3
fl = 2.0 * 27 * 64 * 64 * 4 * 512 * 640
print(f"conv_wide={sys.argv[1] if len(sys.argv) > 1 else 1}: {us:.1f} us  {fl / us / 1e6:.0f} TFLOP/s = {fl / us / 1e6 / 2500:.3f} of the MFMA peak")
