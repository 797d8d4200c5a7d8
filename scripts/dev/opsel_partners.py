#!/usr/bin/env python3
"""Dev (round 4): WHICH partner kernels make `v_pk_fma_f32 ... op_sel:[0,1,0]` fail?  The self-checking `opsel_victim`
(scripts/ubench/liblpo.so) on stream A, one partner kernel family at a time on stream B; wrong results of the three src1-selector forms."""
import ctypes as C
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from wild_deep_mvs_amd import _lib as L, ops, synthetic  # noqa: E402

lpo = C.CDLL(os.path.join(REPO, "scripts", "ubench", "liblpo.so"))
lpo.lpo_opsel_victim.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
g = torch.Generator().manual_seed(1)
dev = "cuda"
mk3 = lambda cin, cout, kind, tr=False: ops.Conv3dLayer.build(torch.randn((cin, cout, 3, 3, 3) if tr else (cout, cin, 3, 3, 3), generator=g) / np.sqrt(27 * cin),
                                                              kind=kind, transposed=tr, device=dev, relu=cout > 1, dtype=torch.float16,
                                                              conv_bias=torch.zeros(1) if cout == 1 else None)
vol = lambda c, d, h, w: (torch.randn(1, d, h, w, c, generator=g) * 0.5).to(torch.float16).to(dev)
x32, x8, x16, x64 = vol(32, 192, 128, 160), vol(8, 192, 128, 160), vol(16, 96, 64, 80), vol(64, 48, 64, 80)
l_c0, l_s2, l_b16, l_b64, l_t2, l_c1 = mk3(32, 8, L.CONV_S1), mk3(8, 16, L.CONV_S2), mk3(16, 16, L.CONV_S1), mk3(64, 64, L.CONV_S1), mk3(16, 8, L.CONV_T2, True), mk3(8, 1, L.CONV_S1)
skip8 = vol(8, 192, 128, 160)
img32 = (torch.randn(5, 128, 160, 32, generator=g) * 0.5).to(torch.float16).to(dev)
img64 = (torch.randn(1, 1024, 1280, 64, generator=g) * 0.5).to(torch.float16).to(dev)
l2_32 = ops.Conv2dLayer.build(torch.randn(32, 32, 3, 3, generator=g) / 17, stride=1, device=dev, relu=True, dtype=torch.float16)
l2_64 = ops.Conv2dLayer.build(torch.randn(64, 64, 3, 3, generator=g) / 24, stride=1, device=dev, relu=True, dtype=torch.float16)
ma, mb = torch.randn(4096, 4096, device=dev, dtype=torch.float16), torch.randn(4096, 4096, device=dev, dtype=torch.float16)
ew = torch.randn(1 << 25, device=dev)
logits = torch.randn(4, 192, 128, 160, device=dev)
dvs = torch.linspace(2, 6, 192, device=dev).view(1, -1).repeat(4, 1)
from wild_deep_mvs_amd.models.MVSNet.model import build_proj_matrices  # noqa: E402
cam = synthetic.make_cameras(1, 5, 512, 640); K = cam["K"].clone(); K[:, :, :2] /= 4
cams = ops.proj_cams_device(build_proj_matrices(K, cam["R"], cam["t"]).to(dev).float().contiguous(), 0)
feats = synthetic.make_features(1, 5, 32, 128, 160, seed=3)
fcl = [ops.to_channels_last(feats[i].to(dev), torch.float16) for i in range(5)]
dv1 = torch.linspace(2, 6, 192, device=dev).view(1, -1)


def quad():
    L.set_tuning("warp_tiled", 0)
    try:
        ops.warp_cost(fcl[0], fcl[1:], cams, dv1, cost=L.COST_VARIANCE, out_dtype=torch.float16)
    finally:
        L.set_tuning("warp_tiled", -1)


partners = [
    ("none", None),
    ("conv0 sweep8 32->8 (MFMA, 70 KB dynamic LDS, 248 VGPRs)", lambda: ops.conv3d(x32, l_c0)),
    ("sweep_s2 8->16 (MFMA, LDS ring)", lambda: ops.conv3d(x8, l_s2)),
    ("brick S1 16->16 (MFMA, dynamic LDS)", lambda: ops.conv3d(x16, l_b16)),
    ("brick S1 64->64 (MFMA, AGPRs)", lambda: ops.conv3d(x64, l_b64)),
    ("t2p8 16->8 (MFMA, static LDS)", lambda: ops.conv3d(x16, l_t2, skip=skip8[:, :, :, :, :] if False else None)),
    ("c1_sweep 8->1 (MFMA)", lambda: ops.conv3d(x8, l_c1)),
    ("conv2d 32->32 (MFMA)", lambda: ops.conv2d(img32, l2_32)),
    ("conv2d wlds 64->64 (MFMA, weights in LDS)", lambda: ops.conv2d(img64, l2_64)),
    ("torch.matmul fp16 4096^3 (hipBLASLt MFMA)", lambda: ma @ mb),
    ("elementwise fp32 (no MFMA, no LDS)", lambda: ew.mul_(1.0001)),
    ("softargmin (VALU + LDS, no MFMA)", lambda: ops.softargmin(logits, dvs, want_conf=True)),
    ("LDS-staged warp, scalar build (VALU + LDS, no MFMA)", lambda: ops.warp_cost(fcl[0], fcl[1:], cams, dv1, cost=L.COST_VARIANCE, out_dtype=torch.float16)),
    ("quad warp (VALU + global taps, no MFMA)", quad),
]
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for name, fn in partners:
    errs = torch.zeros(36, dtype=torch.int32, device=dev)
    for it in range(30):
        if fn is not None:
            with torch.cuda.stream(sb):
                for _ in range(4):
                    fn()
        with torch.cuda.stream(sa):
            assert lpo.lpo_opsel_victim(errs.data_ptr(), 400, sa.cuda_stream) == 0
        torch.cuda.synchronize()
    e = errs.cpu().tolist()
    src1 = {"mul [0,1]": sum(e[8:12]), "add [0,1]": sum(e[16:20]), "fma [0,1,0]": sum(e[24:28])}
    other = sum(e[0:8]) + sum(e[12:16]) + sum(e[20:24]) + sum(e[28:36])
    l48 = e[11] + e[19] + e[27]
    print(f"partner: {name:62s} src1-selector forms wrong: {src1} (in lanes 48-63: {l48}); all other forms: {other}", flush=True)
