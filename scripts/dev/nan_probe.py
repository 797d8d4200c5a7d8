"""Does a NaN input survive the stride-2 conv kernels (brick vs depth sweep)?"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wild_deep_mvs_amd import _lib as L, ops
g = torch.Generator().manual_seed(3)
x = torch.randn(1, 8, 16, 16, 32, generator=g)
x[0, 1, 8, 8, 16] = float("nan")
w = torch.randn(16, 8, 3, 3, 3, generator=g) / 15
for sweep in (0, 2):
    for relu in (False, True):
        for od in (torch.float32, torch.float16):
            L.set_tuning("conv_s2_sweep", sweep)
            layer = ops.Conv3dLayer.build(w, kind=L.CONV_S2, device="cuda", relu=relu, dtype=torch.float16)
            xin = ops.to_channels_last(x.cuda(), torch.float16)
            y = ops.conv3d(xin, layer, out_dtype=od)
            print(f"sweep={sweep} relu={relu} out={od}: input NaNs {int(torch.isnan(xin).sum())}, output NaNs {int(torch.isnan(y).sum())}, infs {int(torch.isinf(y).sum())}, "
                  f"y[0,4,4,8,:4] = {y[0,4,4,8,:4].tolist()}")
L.set_tuning("conv_s2_sweep", 1)
