#!/usr/bin/env python3
"""Dev (round 4): WHERE does a diagnostic build of the packed warp kernel (PSCV_LIB=..., "warp_tiled" = 3) differ from its solo launch
when conv0 runs beside it?  Prints, for the first bad launches: how many values / voxels, histograms over the in-tile pixel (= lane
group), the plane (mod 32 = position in the chunk; parity = wave pair), the channel, and whether whole (tile, plane) rows are hit."""
import collections
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from wild_deep_mvs_amd import _lib as L, ops, synthetic  # noqa: E402
from test_gpu_overlap import Soak, _warp_inputs  # noqa: E402


def main():
    soak = Soak(L, ops)
    fcl, cams, dv = _warp_inputs(ops, synthetic, 5, 32, 128, 160, 192, torch.float16)
    L.set_tuning("warp_tiled", 3)
    run = lambda: ops.warp_cost(fcl[0], fcl[1:], cams, dv, cost=L.COST_VARIANCE, out_dtype=torch.float16)
    ref = run().clone(); torch.cuda.synchronize()
    shown = 0
    for it in range(40):
        with torch.cuda.stream(soak.sb):
            for _ in range(3):
                ops.conv3d(soak.px, soak.player)
        with torch.cuda.stream(soak.sa):
            got = run()
        torch.cuda.synchronize()
        d = got != ref
        n = int(d.sum())
        if not n:
            continue
        idx = d.nonzero().cpu()                       # [n, 5] = b, plane, y, x, c
        vox = idx[:, 1:4].unique(dim=0)
        tiles = torch.stack([vox[:, 0] // 32, vox[:, 1] // 4, vox[:, 2] // 8], 1).unique(dim=0)
        lane_px = collections.Counter(((int(y) % 4), (int(x) % 8)) for _, y, x in vox.tolist())
        plane_in_chunk = collections.Counter(int(p) % 32 for p, _, _ in vox.tolist())
        chans = collections.Counter(int(c) for c in idx[:, 4].tolist())
        per_vox = collections.Counter()
        for b, p, y, x, c in idx.tolist():
            per_vox[(p, y, x)] += 1
        sizes = collections.Counter(per_vox.values())
        err = (got.float() - ref.float()).abs()
        print(f"launch {it}: {n} values in {len(vox)} voxels of {len(tiles)} (chunk, tile) blocks; max abs err {float(err.max()):.3f} (ref max {float(ref.float().abs().max()):.3f})")
        print(f"   channels wrong per voxel -> voxels: {dict(sorted(sizes.items()))}")
        print(f"   in-tile (row, col) -> voxels: {dict(sorted(lane_px.items()))}")
        print(f"   plane mod 32 -> voxels: {dict(sorted(plane_in_chunk.items()))}")
        print(f"   channel -> values: {dict(sorted(chans.items()))}")
        blk = collections.Counter((int(p) // 32, int(y) // 4, int(x) // 8) for p, y, x in vox.tolist())
        print(f"   voxels per hit block (a block holds 1024): {sorted(blk.values(), reverse=True)[:12]}")
        shown += 1
        if shown >= 3:
            break
    if not shown:
        print("no overlapped launch differed")
    L.set_tuning("warp_tiled", -1)


if __name__ == "__main__":
    main()
