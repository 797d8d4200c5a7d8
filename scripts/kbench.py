#!/usr/bin/env python3
"""Per-kernel micro-benchmark at the headline sizes (MVSNet 5-view 128x160x32 features, D=192).
Usage: python scripts/kbench.py [--reps 20] [--only substr] [--dtype bf16|f16]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wild_deep_mvs_amd import _lib as L  # noqa: E402
if os.environ.get("PSCV_LIB"):
    L.LIB_PATH = os.environ["PSCV_LIB"]      # A/B runs against another build of the library
from wild_deep_mvs_amd import ops, synthetic  # noqa: E402


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--dtype", default="bf16")
    args = ap.parse_args()
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[args.dtype]
    dev = "cuda"
    D, h, w = 192, 128, 160
    g = torch.Generator().manual_seed(0)

    def vol(c, s):
        return (torch.randn(1, D // s, h // s, w // s, c, generator=g) * 0.5).to(dt).to(dev)

    rows = []
    # ---- conv layers of the MVSNet regulariser ----
    layers = [("conv0", 32, 8, 0, 1, False), ("conv1", 8, 16, 1, 1, False), ("conv2", 16, 16, 0, 2, False),
              ("conv3", 16, 32, 1, 2, False), ("conv4", 32, 32, 0, 4, False), ("conv5", 32, 64, 1, 4, False),
              ("conv6", 64, 64, 0, 8, False), ("conv7", 64, 32, 2, 8, True), ("conv9", 32, 16, 2, 4, True),
              ("conv11", 16, 8, 2, 2, True), ("prob", 8, 1, 0, 1, False)]
    if args.only == "conv0dc":
        x = vol(32, 1)
        wt = torch.randn(8, 32, 3, 3, 3, generator=g) / (27 * 32) ** 0.5
        layer = ops.Conv3dLayer.build(wt, kind=L.CONV_S1, device=dev, relu=True, dtype=dt)
        out = torch.empty(1, D, h, w, 8, dtype=dt, device=dev)
        for dc in (8, 12, 16, 24, 32, 48, 64, 96):
            L.set_tuning("sweep_dc", dc)
            us = timeit(lambda: ops.conv3d(x, layer, out=out), args.reps)
            print(f"conv0 sweep dc={dc:3d}: {us:8.1f} us")
        L.set_tuning("sweep_dc", 0)
        return
    if args.only == "cvp":
        # the 64-channel layers of CVP-MVSNet (config 4): FeaturePyramid 64 -> 64 at 5 x 1024 x 1280 and the refinement U-Net's
        # 64 -> 64 3-D conv at 4 x 512 x 640
        x2 = (torch.randn(5, 1024, 1280, 64, generator=g) * 0.5).to(dt).to(dev)
        w2 = torch.randn(64, 64, 3, 3, generator=g) / (9 * 64) ** 0.5
        l2 = ops.Conv2dLayer.build(w2, stride=1, device=dev, dtype=dt, leaky=0.1)
        x3 = (torch.randn(1, 4, 512, 640, 64, generator=g) * 0.5).to(dt).to(dev)
        w3 = torch.randn(64, 64, 3, 3, 3, generator=g) / (27 * 64) ** 0.5
        l3 = ops.Conv3dLayer.build(w3, kind=L.CONV_S1, device=dev, relu=True, dtype=dt)
        for (Dv, hv, wv) in ((8, 1024, 1280), (8, 512, 640), (96, 64, 80), (48, 32, 40)):       # 16 -> 16: conv0 / conv0a, MVSNet conv2
            x = (torch.randn(1, Dv, hv, wv, 16, generator=g) * 0.5).to(dt).to(dev)
            wt = torch.randn(16, 16, 3, 3, 3, generator=g) / (27 * 16) ** 0.5
            out = torch.empty(1, Dv, hv, wv, 16, dtype=dt, device=dev)
            res = {}
            for use in (True, False):
                ops.USE_SWEEP_KERNEL, ops.SWEEP16 = use, True
                layer = ops.Conv3dLayer.build(wt, kind=L.CONV_S1, device=dev, relu=True, dtype=dt)
                ops.USE_SWEEP_KERNEL, ops.SWEEP16 = True, False
                res[use] = timeit(lambda: ops.conv3d(x, layer, out=out), args.reps)
            gb = Dv * hv * wv * 64 / 1e9
            print(f"conv3d 16->16 @ {Dv}x{hv}x{wv}: sweep {res[True]:8.1f} us ({gb / res[True] * 1e6:6.0f} GB/s)   brick {res[False]:8.1f} us")
            if args.reps > 20:
                ops.SWEEP16 = True
                layer = ops.Conv3dLayer.build(wt, kind=L.CONV_S1, device=dev, relu=True, dtype=dt)
                ops.SWEEP16 = False
                for pd in (1, 2, 3):
                    L.set_tuning("sweepc_pd", pd)
                    row = []
                    for slots in (512, 768, 1024, 2048):
                        L.set_tuning("sweepc_slots", slots)
                        row.append(f"{slots}: {timeit(lambda: ops.conv3d(x, layer, out=out), args.reps):6.1f}")
                    print(f"    prefetch distance {pd}  " + "  ".join(row))
                L.set_tuning("sweepc_pd", 0); L.set_tuning("sweepc_slots", 0)
        t2 = timeit(lambda: ops.conv2d(x2, l2), args.reps)
        t3 = timeit(lambda: ops.conv3d(x3, l3), args.reps)
        f2 = 5 * 1024 * 1280 * 64 * 64 * 9 * 2 / 1e12
        f3 = 4 * 512 * 640 * 64 * 64 * 27 * 2 / 1e12
        print(f"conv2d 64->64 @5x1024x1280 {t2:8.1f} us ({f2 / t2 * 1e6:6.0f} TFLOP/s)   "
              f"conv3d 64->64 @4x512x640 {t3:8.1f} us ({f3 / t3 * 1e6:6.0f} TFLOP/s)")
        return
    if args.only == "vis":
        # the Vis-MVSNet U-Net's full-resolution layers (8 -> 8 with residual, 16 -> 8 after the concat): narrow depth-sweep
        # kernel (default) against the generic brick kernel, at stage-1 sizes of BASELINE configurations 3 and 5
        for (Dv, hv, wv) in ((192, 64, 80), (32, 128, 160), (16, 256, 320), (256, 144, 200), (32, 288, 400), (16, 576, 800)):
            for ci in (8, 16):
                x = (torch.randn(1, Dv, hv, wv, 16, generator=g) * 0.5).to(dt).to(dev)
                sk = (torch.randn(1, Dv, hv, wv, 8, generator=g) * 0.5).to(dt).to(dev)
                wt = torch.randn(8, ci, 3, 3, 3, generator=g) / (27 * ci) ** 0.5
                out = torch.empty(1, Dv, hv, wv, 8, dtype=dt, device=dev)
                vox = Dv * hv * wv
                gb = vox * (ci + 8 + 8) * 2 / 1e9
                res = {}
                for use in (True, False):
                    ops.USE_SWEEP_KERNEL = use
                    layer = ops.Conv3dLayer.build(wt, kind=L.CONV_S1, device=dev, relu_post=True, dtype=dt)
                    ops.USE_SWEEP_KERNEL = True
                    res[use] = timeit(lambda: ops.conv3d(x, layer, skip=sk, out=out), args.reps)
                print(f"vis {ci:2d}->8 + skip @ {Dv}x{hv}x{wv}: sweep {res[True]:8.1f} us ({gb / res[True] * 1e6:6.0f} GB/s)   "
                      f"brick {res[False]:8.1f} us ({gb / res[False] * 1e6:6.0f} GB/s)")
                if True:
                    layer = ops.Conv3dLayer.build(wt, kind=L.CONV_S1, device=dev, relu_post=True, dtype=dt)
                    for pd in ((1, 2, 3) if args.reps > 20 else ()):
                        L.set_tuning("sweepc_pd", pd)
                        row = []
                        for slots in (512, 768, 1024, 2048):
                            L.set_tuning("sweepc_slots", slots)
                            row.append(f"{slots}: {timeit(lambda: ops.conv3d(x, layer, skip=sk, out=out), args.reps):6.1f}")
                        print(f"    prefetch distance {pd}, us by resident-workgroup target  " + "  ".join(row))
                    L.set_tuning("sweepc_slots", 0)
                    L.set_tuning("sweepc_pd", 0)
        return
    for name, ci, co, kind, s, skip in layers:
        if args.only and args.only not in name:
            continue
        x = vol(ci, s)
        transposed = kind == L.CONV_T2
        wshape = (ci, co, 3, 3, 3) if transposed else (co, ci, 3, 3, 3)
        wt = torch.randn(wshape, generator=g) / (27 * ci) ** 0.5
        try:
            layer = ops.Conv3dLayer.build(wt, kind=kind, transposed=transposed, device=dev, relu=True, dtype=dt)
        except TypeError:
            layer = ops.Conv3dLayer.build(wt, kind=kind, transposed=transposed, device=dev, relu=True)
        Do, Ho, Wo = ops.conv_out_shape(kind, *x.shape[1:4])
        sk = (torch.randn(1, Do, Ho, Wo, co, generator=g)).to(dt).to(dev) if skip else None
        out = torch.empty(1, Do, Ho, Wo, co, dtype=torch.float32 if name == "prob" else dt, device=dev)
        us = timeit(lambda: ops.conv3d(x, layer, skip=sk, out=out), args.reps)
        nbytes = x.numel() * 2 + out.numel() * out.element_size() + (sk.numel() * 2 if skip else 0)
        flops = 2 * 27 * ci * co * (Do * Ho * Wo if kind != L.CONV_T2 else x.shape[1] * x.shape[2] * x.shape[3])
        rows.append((f"conv3d {name} {ci}->{co} k{kind}", us, nbytes / us / 1e3, flops / us / 1e6))
    # ---- warp + cost ----
    if not args.only or "warp" in args.only:
        feats = synthetic.make_features(1, 5, 32, h, w, seed=1)
        fcl = [ops.to_channels_last(feats[i].to(dev), dt) for i in range(5)]
        cams = synthetic.make_cameras(1, 5, 512, 640)
        from wild_deep_mvs_amd.models.MVSNet.model import build_proj_matrices
        Ks = cams["K"].clone(); Ks[:, :, :2] /= 4
        proj = build_proj_matrices(Ks, cams["R"], cams["t"]).to(dev)
        dv = torch.linspace(2.0, 6.0, D).view(1, D).to(dev)
        cm = ops.proj_cams([proj[:, i] for i in range(1, 5)], proj[:, 0])
        out = torch.empty(1, D, h, w, 32, dtype=dt, device=dev)
        nbytes = 5 * 32 * h * w * 2 + out.numel() * 2
        for ppd in (16,):
            L.set_tuning("warp_lpv", 0); L.set_tuning("warp_tiled", 1); L.set_tuning("warp_ppd", ppd)
            us = timeit(lambda: ops.warp_cost(fcl[0], fcl[1:], cm, dv, cost=L.COST_VARIANCE, out=out), args.reps)
            rows.append((f"warp_cost variance TILED ppd={ppd}", us, nbytes / us / 1e3, 0))
            L.set_tuning("warp_tiled", -1)
        for lpv in (4, 2, 1):
            for ppd in (4, 8, 16):
                L.set_tuning("warp_lpv", lpv); L.set_tuning("warp_ppd", ppd)
                us = timeit(lambda: ops.warp_cost(fcl[0], fcl[1:], cm, dv, cost=L.COST_VARIANCE, out=out), args.reps)
                rows.append((f"warp_cost variance lpv={lpv} ppd={ppd}", us, nbytes / us / 1e3, 0))
        L.set_tuning("warp_lpv", 0); L.set_tuning("warp_ppd", 0)
    # ---- softargmin ----
    if not args.only or "soft" in args.only:
        logits = torch.randn(1, D, h, w, generator=g).to(dev)
        dv = torch.linspace(2.0, 6.0, D).view(1, D).to(dev)
        us = timeit(lambda: ops.softargmin(logits, dv, want_conf=True), args.reps)
        rows.append(("softargmin fp32 logits", us, logits.numel() * 4 / us / 1e3, 0))
    for name, us, gbs, tf in rows:
        print(f"{name:44s} {us:10.1f} us  {gbs:8.1f} GB/s(alg)  {tf:8.2f} TFLOP/s")


if __name__ == "__main__":
    main()
