#!/usr/bin/env python3
"""Dev (rounds 5-6): interleaved A/B timing of the LDS-staged warp kernel under the values of the `warp_tile` knob (0 = adaptive split of a
chunk whose boxes do not fit into halves and, round 6, quarters; 3 = halves only (round 5's kernel); 1 = no split) and of the
direct-gather kernel, on both camera rigs, with a bit comparison of every arm against the direct-gather kernel; several rounds,
alternating, so that clock ramps and box-to-box differences cancel.  Usage: python scripts/dev/warp_ab.py [rounds]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wild_deep_mvs_amd import _lib as L, ops, synthetic  # noqa: E402
from wild_deep_mvs_amd.models.MVSNet.model import build_proj_matrices  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    dev, dt, V, D, h, w = "cuda", torch.float16, 5, 192, 128, 160
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rig in ("probe", "dtu"):
        feats = synthetic.make_features(1, V, 32, h, w, seed=1)
        fcl = [ops.to_channels_last(feats[i].to(dev), dt) for i in range(V)]
        cams = synthetic.make_cameras(1, V, 512, 640, rig=rig)
        Ks = cams["K"].clone(); Ks[:, :, :2] /= 4
        proj = build_proj_matrices(Ks, cams["R"], cams["t"]).to(dev)
        dv = torch.linspace(float(cams["depth_min"][0, 0]), float(cams["depth_max"][0, 0]), D).view(1, D).to(dev)
        cm = ops.proj_cams([proj[:, i] for i in range(1, V)], proj[:, 0])
        out = torch.empty(1, D, h, w, 32, dtype=dt, device=dev)
        cfgs = {"split (halves + quarters)": dict(warp_tiled=1, warp_tile=0), "split (halves only)": dict(warp_tiled=1, warp_tile=3),
                "no split": dict(warp_tiled=1, warp_tile=1), "direct gather": dict(warp_tiled=0, warp_tile=0),
                "lane owner, split": dict(warp_tiled=4, warp_tile=2), "lane owner, no split": dict(warp_tiled=4, warp_tile=0)}
        acc = {k: [] for k in cfgs}
        L.set_tuning("warp_tiled", 0); L.set_tuning("warp_tile", 0)
        ref = ops.warp_cost(fcl[0], fcl[1:], cm, dv, cost=L.COST_VARIANCE, out_dtype=dt).clone()
        same = {}
        for name, kn in cfgs.items():
            for k, v in kn.items():
                L.set_tuning(k, v)
            same[name] = bool(torch.equal(ops.warp_cost(fcl[0], fcl[1:], cm, dv, cost=L.COST_VARIANCE, out_dtype=dt), ref))
        for r in range(rounds + 1):
            for name, kn in cfgs.items():
                for k, v in kn.items():
                    L.set_tuning(k, v)
                for _ in range(3):
                    ops.warp_cost(fcl[0], fcl[1:], cm, dv, cost=L.COST_VARIANCE, out=out)
                e0.record()
                for _ in range(20):
                    ops.warp_cost(fcl[0], fcl[1:], cm, dv, cost=L.COST_VARIANCE, out=out)
                e1.record()
                torch.cuda.synchronize()
                if r:                                   # round 0 = warm-up
                    acc[name].append(e0.elapsed_time(e1) / 20 * 1e3)
        L.set_tuning("warp_tiled", -1); L.set_tuning("warp_tile", 0)
        print(f"{rig}: " + "; ".join(f"{k} {sorted(v)[len(v) // 2]:.1f} us (min {min(v):.1f}, bits = direct gather: {same[k]})" for k, v in acc.items()), flush=True)


if __name__ == "__main__":
    main()
