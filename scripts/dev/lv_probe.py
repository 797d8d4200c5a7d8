#!/usr/bin/env python3
"""Dev (round 4): the lane-owns-voxel warp kernel ("warp_tiled" = 4, warp_cost_lv.hip) against the quad-owner kernel (default) and the
direct-gather kernel (0) at the headline size: stored bits equal? kernel time (events, stand-alone launches in a loop)? both rigs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from wild_deep_mvs_amd import _lib as L, ops  # noqa: E402


def time_us(run, steps=40, warm=10):
    for _ in range(warm):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(steps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


def main():
    dev = torch.device("cuda", 0)
    quick = "--quick" in sys.argv
    for a_ in sys.argv:
        if a_.startswith("--ppd="):
            L.set_tuning("warp_ppd", int(a_[6:]))
    for rig in (("probe",) if quick else ("probe", "dtu")):
        for dtype in ((torch.float16,) if quick else (torch.float16, torch.bfloat16)):
            from wild_deep_mvs_amd import synthetic
            D = bench.D
            cm = synthetic.make_cameras(1, bench.V, bench.IMG_H, bench.IMG_W, rig=rig)
            Ks = cm["K"].clone()
            Ks[:, :, :2] /= 4
            proj_d = bench.build_proj_matrices(Ks, cm["R"], cm["t"]).to(dev)
            steps = torch.arange(D, dtype=torch.float32).view(1, -1)
            dv_d = (cm["depth_min"][:, :1] + (cm["depth_max"][:, :1] - cm["depth_min"][:, :1]) / (D - 1) * steps).to(dev).contiguous()
            feats = synthetic.make_features(1, bench.V, bench.C, bench.h, bench.w, seed=7)
            fcl = [ops.to_channels_last(feats[i].to(dev), dtype) for i in range(bench.V)]
            cams = ops.proj_cams_device(proj_d.float().contiguous(), 0)
            outs, times = {}, {}
            for tiled in ((1, 4, 0) if not quick else (1, 4)):
                L.set_tuning("warp_tiled", tiled)
                try:
                    out = torch.empty((1, D, bench.h, bench.w, 32), dtype=dtype, device=dev)
                    run = lambda: ops.warp_cost(fcl[0], fcl[1:], cams, dv_d, cost=L.COST_VARIANCE, out=out)  # noqa: E731
                    import ctypes
                    hist = torch.zeros(16, dtype=torch.int32, device=dev)
                    fn = L.lib().pscv_debug_wl_mode_hist
                    fn.argtypes, fn.restype = [ctypes.c_void_p], None
                    fn(hist.data_ptr())
                    run(); torch.cuda.synchronize()
                    fn(None)
                    print(f"  warp_tiled={tiled} modes per view [DIRECT, GEN, FAST, ZERO]:", hist.view(4, 4).cpu().tolist(), flush=True)
                    outs[tiled] = out.clone()
                    times[tiled] = time_us(run, *( (5, 2) if quick else (40, 10)))
                finally:
                    L.set_tuning("warp_tiled", -1)
            if "--ablate" in sys.argv:
                outb = torch.empty((1, D, bench.h, bench.w, 32), dtype=dtype, device=dev)
                for name, v in (("full", 0), ("no stores", 8), ("no taps / blend (staging + stores only)", 9), ("no taps, no stores (box + staging only)", 10)):
                    L.set_tuning("warp_tiled", 4); L.set_tuning("warp_tile", v)
                    ts = [time_us(lambda: ops.warp_cost(fcl[0], fcl[1:], cams, dv_d, cost=L.COST_VARIANCE, out=outb), 30, 5) for _ in range(5)]
                    L.set_tuning("warp_tiled", -1); L.set_tuning("warp_tile", 0)
                    print(f"  lane-owner ablation, {name}: median {sorted(ts)[2]:.1f} us", flush=True)
            if "--general" in sys.argv:
                L.set_tuning("warp_tiled", 4); L.set_tuning("warp_tile", 7)
                out = torch.empty((1, D, bench.h, bench.w, 32), dtype=dtype, device=dev)
                ops.warp_cost(fcl[0], fcl[1:], cams, dv_d, cost=L.COST_VARIANCE, out=out)
                torch.cuda.synchronize()
                L.set_tuning("warp_tiled", -1); L.set_tuning("warp_tile", 0)
                print("  every block on the general path: values differing vs quad-owner",
                      int((out.view(torch.int16) != outs[1].view(torch.int16)).sum()), flush=True)
            if "--ab" in sys.argv:       # interleaved A/B: medians of 7 rounds x 30 launches
                rounds = {1: [], 4: []}
                outb = torch.empty((1, D, bench.h, bench.w, 32), dtype=dtype, device=dev)
                for _ in range(7):
                    for tiled in (1, 4):
                        L.set_tuning("warp_tiled", tiled)
                        rounds[tiled].append(time_us(lambda: ops.warp_cost(fcl[0], fcl[1:], cams, dv_d, cost=L.COST_VARIANCE, out=outb), 30, 5))
                        L.set_tuning("warp_tiled", -1)
                med = {k: sorted(v)[len(v) // 2] for k, v in rounds.items()}
                print(f"  A/B interleaved, median us: quad-owner {med[1]:.1f} (min {min(rounds[1]):.1f}), lane-owner {med[4]:.1f} (min {min(rounds[4]):.1f})", flush=True)
            ne1 = int((outs[4].view(torch.int16) != outs[1].view(torch.int16)).sum())
            ne0 = int((outs[4].view(torch.int16) != outs[0].view(torch.int16)).sum()) if 0 in outs else -1
            times.setdefault(0, 0.0)
            if ne1:
                a4, a1 = outs[4][0].float(), outs[1][0].float()          # [D, h, w, C]
                bad = (outs[4][0].view(torch.int16) != outs[1][0].view(torch.int16))
                idx = bad.nonzero()
                print("  max abs diff", float((a4 - a1).abs().max()), "max |value|", float(a1.abs().max()))
                print("  by channel chunk:", torch.bincount(idx[:, 3] // 4, minlength=8).tolist())
                print("  by plane % 32:", torch.bincount(idx[:, 0] % 32, minlength=32).tolist())
                tiles = (idx[:, 1] // 4) * 1000 + idx[:, 2] // 8
                ut, cnt = torch.unique(tiles * 8 + idx[:, 0] // 32, return_counts=True)
                print("  blocks with differences:", ut.numel(), "of", (bench.h // 4) * (bench.w // 8) * 6, "; per-block counts (first 12):", cnt[:12].tolist())
                k = idx[0].tolist()
                print("  first:", k, float(a4[tuple(k)]), float(a1[tuple(k)]))
            print(f"{rig} {str(dtype)[6:]}: us direct {times[0]:.1f} quad-owner {times[1]:.1f} lane-owner {times[4]:.1f}; "
                  f"values differing lane-owner vs quad-owner {ne1}, vs direct {ne0} of {outs[4].numel()}", flush=True)


if __name__ == "__main__":
    main()
