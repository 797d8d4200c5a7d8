"""Phase stamps of conv2d_wlds_kernel (-DPSCV_PROFILE build: bash scripts/dev/ab_build.sh c2prof conv2d.hip -DPSCV_PROFILE)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wild_deep_mvs_amd import _lib as L
L.LIB_PATH = os.environ["PSCV_LIB"]
from wild_deep_mvs_amd import ops
g = torch.Generator().manual_seed(0)
B, H, W = 5, 1024, 1280
for co in (64, 32):
    w = torch.randn(co, 64, 3, 3, generator=g) / 24
    layer = ops.Conv2dLayer.build(w, stride=1, device="cuda", leaky=0.1, dtype=torch.float16)
    x = (torch.randn(B, H, W, 64, generator=g) * 0.5).to(torch.float16).cuda()
    L.set_tuning("conv2d_wlds", 2)
    for _ in range(3):
        ops.conv2d(x, layer)
    torch.cuda.synchronize()
    nb = 256
    raw = (ctypes.c_uint * (nb * 16))()
    L.lib().pscv_debug_prof_c2w(raw, nb)
    r = np.frombuffer(raw, dtype=np.uint32).reshape(nb, 16).astype(np.int64)
    tiles = B * (H // 8) * (W // 32) / 256
    names = ["brick landed+written", "barrier", "fetch issued", "k-loop", "epilogue", "end barrier"]
    print(f"64->{co}: {tiles:.0f} tiles per workgroup; cycles per tile (wave 0 of each workgroup, mean over workgroups):")
    for i, nme in enumerate(names):
        print(f"   {nme:22s} {r[:, i].mean() / tiles:8.0f}")
    print(f"   total                  {(r[:, 7] - r[:, 6]).mean() / tiles:8.0f}")
