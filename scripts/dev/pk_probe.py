#!/usr/bin/env python3
"""Dev (round 4): where and when does the PACKED build of the LDS-staged warp kernel ("warp_tiled" = 3) differ from the scalar build?
(1) solo launches (nothing else on the GPU), small and headline size; (2) beside conv0 (tests/test_gpu_overlap.py's harness);
(3) the stand-alone victim of scripts/ubench/lds_pk_overlap.hip (liblpo.so) beside the engine's REAL conv0."""
import ctypes as C
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from wild_deep_mvs_amd import _lib as L, ops, synthetic  # noqa: E402
from test_gpu_overlap import Soak, _warp_inputs  # noqa: E402


def where(ref, got, tag):
    d = (ref != got)
    n = int(d.sum())
    if not n:
        return
    idx = d.nonzero()[:4000].cpu()          # [n, 5] = b, plane, y, x, c
    import collections
    lanes = collections.Counter()
    for b, p, y, x, c in idx.tolist():
        lanes[(y % 4, x % 8)] += 1
    vox = {(b, p, y, x) for b, p, y, x, c in idx.tolist()}
    chans = collections.Counter(c for *_, c in idx.tolist())
    print(f"   {tag}: {n} values in {len(vox)}+ voxels; in-tile (row, col) histogram {dict(sorted(lanes.items()))}; channels {dict(sorted(chans.items()))}")
    b, p, y, x, c = idx[0].tolist()
    print(f"   first: b={b} plane={p} y={y} x={x} c={c} ref={float(ref[b, p, y, x, c]):.6f} got={float(got[b, p, y, x, c]):.6f}")


def main():
    soak = Soak(L, ops)
    for (h, w, D, B) in ((64, 80, 24, 2), (128, 160, 192, 1)):
        fcl, cams, dv = _warp_inputs(ops, synthetic, 5, 32, h, w, D, torch.float16)
        if B == 2:
            fcl = [f.repeat(2, 1, 1, 1).contiguous() for f in fcl]; cams = cams.repeat(1, 2, 1).contiguous(); dv = dv.repeat(2, 1).contiguous()
        run = lambda: ops.warp_cost(fcl[0], fcl[1:], cams, dv, cost=L.COST_VARIANCE, out_dtype=torch.float16)
        L.set_tuning("warp_tiled", 1)
        ref = run().clone(); torch.cuda.synchronize()
        again = run(); torch.cuda.synchronize()
        print(f"[{h}x{w} D={D} B={B}] scalar build repeat equal: {torch.equal(ref, again)}")
        L.set_tuning("warp_tiled", 3)
        bad = 0
        for it in range(60):
            got = run(); torch.cuda.synchronize()
            if not torch.equal(got, ref):
                bad += 1
                if bad <= 3:
                    where(ref, got, f"solo launch {it}")
        print(f"[{h}x{w} D={D} B={B}] PACKED build, SOLO: {bad} of 60 launches differ from the scalar build")
        # solo, but right behind a burst of conv0 launches on the SAME stream (no overlap; clocks / power state of an MFMA phase)
        bad = 0
        for it in range(40):
            for _ in range(6):
                ops.conv3d(soak.px, soak.player)
            got = run(); torch.cuda.synchronize()
            if not torch.equal(got, ref):
                bad += 1
                if bad <= 2:
                    where(ref, got, f"after-conv0 launch {it}")
        print(f"[{h}x{w} D={D} B={B}] PACKED build, same stream right behind 6 conv0 launches: {bad} of 40 differ")
        b2, _, _ = soak.run(f"PACKED build {h}x{w} beside conv0", run, launches=80)
        L.set_tuning("warp_tiled", 1)
        b1, _, _ = soak.run(f"scalar build {h}x{w} beside conv0", run, launches=80)
    L.set_tuning("warp_tiled", -1)
    # the stand-alone victim beside the REAL conv0
    lpo = C.CDLL(os.environ.get("LPO_LIB") or os.path.join(REPO, "scripts", "ubench", "liblpo.so"))
    lpo.lpo_victim.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lpo.lpo_out_floats.restype = C.c_long
    n = lpo.lpo_out_floats()
    for pk in (1, 0):
        def vic():
            out = torch.empty(n, dtype=torch.float32, device="cuda")
            rc = lpo.lpo_victim(out.data_ptr(), 96, pk, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
            return out
        soak.run(f"stand-alone victim (pk={pk}) beside the real conv0", vic, launches=120)


if __name__ == "__main__":
    main()
