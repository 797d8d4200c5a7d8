#!/usr/bin/env python3
"""A/B of the 32 -> 8 depth sweep: plane-pair kernel (sweep_kdm = 0) against the kd-in-rows kernel (1: three workgroups per CU
as the chunking target, 2: four).  Checks both against a torch fp32 conv3d of the same 16-bit operands, then times them.
The ablation lines at the end mean something only with a library built with -DPSCV_ABLATE (bash scripts/dev/ab_build.sh abl
conv3d_sweep.hip -DPSCV_ABLATE; PSCV_LIB=scripts/dev/libpscv_abl.so): "fuse_c0" bits 1 / 2 / 4 switch off the plane fetch, the
stores, the LDS reads + MFMAs of the kd-in-rows kernel.
Usage: python scripts/dev/kdm_bench.py [--reps 50] [--dtype f16|bf16] [--small]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wild_deep_mvs_amd import _lib as L  # noqa: E402
from wild_deep_mvs_amd import ops  # noqa: E402


def timeit(fn, reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--small", action="store_true")
    args = ap.parse_args()
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[args.dtype]
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    shapes = [(1, 20, 19, 37), (2, 7, 8, 16)] if args.small else [(1, 20, 19, 37), (1, 192, 128, 160)]
    wt = torch.randn(8, 32, 3, 3, 3, generator=g) / (27 * 32) ** 0.5
    layer = ops.Conv3dLayer.build(wt, kind=L.CONV_S1, device=dev, relu=True, dtype=dt)
    w16 = wt.to(dt).float().to(dev)
    for (B, D, h, w) in shapes:
        x = (torch.randn(B, D, h, w, 32, generator=g) * 0.5).to(dt).to(dev)
        ref = F.relu(F.conv3d(x.float().permute(0, 4, 1, 2, 3), w16, padding=1)).permute(0, 2, 3, 4, 1)
        for knob, pd in ((0, 1), (1, 1), (2, 1), (1, 2), (2, 2)):
            L.set_tuning("sweep_kdm", knob)
            L.set_tuning("sweep_kdm_pd", pd)
            out = torch.full((B, D, h, w, 8), float("nan"), dtype=dt, device=dev)
            ops.conv3d(x, layer, out=out)
            torch.cuda.synchronize()
            err = (out.float() - ref).abs().max().item()
            rel = ((out.float() - ref).norm() / ref.norm()).item()
            us = timeit(lambda: ops.conv3d(x, layer, out=out), args.reps)
            print(f"shape {B}x{D}x{h}x{w} sweep_kdm={knob} pd={pd}: max abs err {err:.3e} rel-L2 {rel:.3e} nan {int(torch.isnan(out).sum())}  {us:8.1f} us", flush=True)
        L.set_tuning("sweep_kdm_pd", 1)
        for dc in (16, 24, 32, 48, 64, 96):
            L.set_tuning("sweep_kdm", 1)
            L.set_tuning("sweep_dc", dc)
            us = timeit(lambda: ops.conv3d(x, layer, out=out), args.reps)
            print(f"   kdm dc={dc}: {us:8.1f} us", flush=True)
        L.set_tuning("sweep_dc", 0)
        for pd in (1, 2):
          L.set_tuning("sweep_kdm_pd", pd)
          for fl in (0, 1, 2, 4, 3, 5, 6, 7):
            L.set_tuning("fuse_c0", fl)
            us = timeit(lambda: ops.conv3d(x, layer, out=out), args.reps)
            print(f"   kdm pd={pd} ablation no-fetch={fl & 1} no-store={(fl >> 1) & 1} no-mfma={(fl >> 2) & 1}: {us:8.1f} us", flush=True)
        L.set_tuning("fuse_c0", 0)
    L.set_tuning("sweep_kdm", 0)


if __name__ == "__main__":
    main()
