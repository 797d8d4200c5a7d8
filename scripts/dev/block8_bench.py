"""Fused residual block (pscv_conv3d_block8) against its two depth-sweep launches at the Vis stage shapes of configurations 3 / 5."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from wild_deep_mvs_amd import _lib as L, ops
g = torch.Generator().manual_seed(0)
mk = lambda relu, post: ops.Conv3dLayer.build(torch.randn(8, 8, 3, 3, 3, generator=g) / np.sqrt(216), kind=L.CONV_S1P8, device="cuda", dtype=torch.float16,
                                             bn=(torch.ones(8), torch.zeros(8), torch.zeros(8), torch.ones(8)), relu=relu, relu_post=post)
l1, l2 = mk(True, False), mk(False, True)
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for shp in ([(16, 576, 800)] if 'B8_SLOTS' in os.environ else [(16, 576, 800), (32, 288, 400), (256, 144, 200), (16, 256, 320), (32, 128, 160), (192, 64, 80)]):
    x = torch.randn((1,) + shp + (8,), generator=g).to(torch.float16).cuda()
    two = t(lambda: ops.conv3d(ops.conv3d(x, l1), l2, skip=x))
    res = []
    for slots in [int(v, 0) for v in os.environ.get('B8_SLOTS', '0,512,1024,1536').split(',')]:
        L.set_tuning("block8_slots", slots)
        res.append((slots, t(lambda: ops.conv3d_block8(x, l1, l2))))
    L.set_tuning("block8_slots", 0)
    vox = np.prod(shp)
    print(f"{shp}: two launches {two:7.1f} us; fused " + ", ".join(f"[{s}] {u:6.1f}" for s, u in res) + f"  -> {vox * 32 / (min(u for _, u in res) * 1e-6) / 1e12:.2f} TB/s algorithmic")
