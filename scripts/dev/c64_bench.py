"""64-channel 3x3 layers of CVP-MVSNet's FeaturePyramid at configuration-4 sizes: conv2d_kernel (weights per wave from L2) against
conv2d_wlds_kernel (weights resident in LDS, persistent workgroups)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wild_deep_mvs_amd import _lib as L, ops
g = torch.Generator().manual_seed(0)
for (B, H, W) in [(5, 1024, 1280), (5, 512, 640), (5, 256, 320), (5, 128, 160)]:
    for co in (64, 32):
        w = torch.randn(co, 64, 3, 3, generator=g) / 24
        layer = ops.Conv2dLayer.build(w, stride=1, device="cuda", leaky=0.1, dtype=torch.float16)
        x = (torch.randn(B, H, W, 64, generator=g) * 0.5).to(torch.float16).cuda()
        outs = {}
        for knob in (0, 2):
            L.set_tuning("conv2d_wlds", knob)
            for _ in range(2):
                y = ops.conv2d(x, layer)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                y = ops.conv2d(x, layer)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 5 * 1e3
            fl = 2.0 * 9 * 64 * co * B * H * W
            by = B * H * W * (64 + co) * 2
            outs[knob] = y
            print(f"{B}x{H}x{W} 64->{co} wlds={knob}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s ({fl / us / 1e6 / 2500:.3f} of MFMA peak)  {by / us / 1e3:7.0f} GB/s")
        L.set_tuning("conv2d_wlds", 1)
        print("   equal:", bool(torch.equal(outs[0], outs[2])))
