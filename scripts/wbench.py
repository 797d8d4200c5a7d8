#!/usr/bin/env python3
"""Warp + cost kernel micro-benchmark at the headline size (MVSNet 5-view 128x160x32 features, D=192): the quad-mapped
direct kernel against the LDS-staged kernel, both 16-bit formats, and a bit-comparison of their outputs.
Usage: python scripts/wbench.py [--reps 20] [--dtype f16|bf16] [--only tiled|q2] [--ppd N] [--scene dtu]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wild_deep_mvs_amd import _lib as L  # noqa: E402
if os.environ.get("PSCV_LIB"):
    L.LIB_PATH = os.environ["PSCV_LIB"]      # A/B runs against another build of the library
from wild_deep_mvs_amd import ops, synthetic  # noqa: E402
from wild_deep_mvs_amd.models.MVSNet.model import build_proj_matrices  # noqa: E402


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
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--only", default="")
    ap.add_argument("--ppd", type=int, nargs="*", default=[0])
    ap.add_argument("--views", type=int, default=5)
    ap.add_argument("--baseline-scale", type=float, default=1.0)
    ap.add_argument("--rig", default="probe", help="probe | dtu (synthetic.make_cameras)")
    ap.add_argument("--variants", type=int, nargs="*", default=[0], help='values of the "warp_tile" knob to time on the LDS-staged kernel '
                                                                         "(0 = default pipelined sweep, 1 = the round-2 loop)")
    args = ap.parse_args()
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[args.dtype]
    dev = "cuda"
    V, D, h, w = args.views, 192, 128, 160
    feats = synthetic.make_features(1, V, 32, h, w, seed=1)
    fcl = [ops.to_channels_last(feats[i].to(dev), dt) for i in range(V)]
    cams = synthetic.make_cameras(1, V, 512, 640, rig=args.rig)
    cams["t"] = cams["t"] * args.baseline_scale
    Ks = cams["K"].clone(); Ks[:, :, :2] /= 4
    proj = build_proj_matrices(Ks, cams["R"], cams["t"]).to(dev)
    dv = torch.linspace(float(cams["depth_min"][0, 0]), float(cams["depth_max"][0, 0]), D).view(1, D).to(dev)
    cm = ops.proj_cams([proj[:, i] for i in range(1, V)], proj[:, 0])
    nbytes = V * 32 * h * w * 2 + D * h * w * 32 * 2
    outs = {}
    for name, tiled in (("q2", 0), ("tiled", 1)):
        if args.only and args.only != name:
            continue
        for ppd in args.ppd:
            for variant in (args.variants if tiled else [0]):
                out = torch.empty(1, D, h, w, 32, dtype=dt, device=dev)
                L.set_tuning("warp_tiled", tiled); L.set_tuning("warp_ppd", ppd); L.set_tuning("warp_tile", variant)
                hist = torch.zeros(16, dtype=torch.int32, device=dev)
                try:
                    if tiled:
                        import ctypes
                        fn = L.lib().pscv_debug_wl_mode_hist
                        fn.argtypes, fn.restype = [ctypes.c_void_p], None
                        fn(hist.data_ptr())
                        ops.warp_cost(fcl[0], fcl[1:], cm, dv, cost=L.COST_VARIANCE, out=out)
                        torch.cuda.synchronize()
                        fn(None)
                        print("   staging modes per view [DIRECT, GEN, FAST, ZERO]:", hist.view(4, 4).tolist())
                    us = timeit(lambda: ops.warp_cost(fcl[0], fcl[1:], cm, dv, cost=L.COST_VARIANCE, out=out), args.reps)
                finally:
                    L.set_tuning("warp_tiled", -1); L.set_tuning("warp_ppd", 0); L.set_tuning("warp_tile", 0)
                if tiled and variant != args.variants[0]:
                    ne = (outs[name].view(torch.int16) != out.view(torch.int16)).sum().item()
                    print(f"   variant {variant} vs {args.variants[0]}: {ne} stored values differ")
                else:
                    outs[name] = out
                print(f"warp_cost variance {args.dtype} {name:6s} ppd={ppd:2d} variant={variant}: {us:8.1f} us  {nbytes / us / 1e3:7.1f} GB/s (algorithmic)"
                      f"  = {nbytes / us / 1e3 / 8000:.3f} of 8 TB/s")
    if len(outs) == 2:
        a, b = outs["q2"].float(), outs["tiled"].float()
        ne = (outs["q2"].view(torch.int16) != outs["tiled"].view(torch.int16)).sum().item()
        print(f"q2 vs tiled: {ne} of {a.numel()} stored values differ, max abs {float((a - b).abs().max()):.3e}")


if __name__ == "__main__":
    main()
