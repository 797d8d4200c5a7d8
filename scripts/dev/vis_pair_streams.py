#!/usr/bin/env python3
"""Dev (round 4): Vis-MVSNet configurations 3 and 5, the per-view pair passes of a stage on 1 / 2 / 3 / 4 HIP streams (parallel branches
of the forward's replayed graph): ms per forward and equality of the depth map with the one-stream run."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import run_configs as RC  # noqa: E402
from wild_deep_mvs_amd import synthetic  # noqa: E402


def timed(call, reps=7):
    call(); call(); call(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); out = call(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3, out


for cid in (3, 5):
    cfg = RC.CONFIGS[cid]
    net = RC.build(cfg["arch"])
    scene = {k: v.cuda() for k, v in synthetic.make_scene(1, cfg["V"], cfg["H"], cfg["W"], seed=cid).items()}
    call = lambda: net(scene["imgs"], scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"], **cfg["kw"])
    stages = (net.model.stage1, net.model.stage2, net.model.stage3)
    with torch.no_grad():
        base_ms, base = timed(call)
        print(f"config {cid}: default (PAIR_STREAMS 1, PAIR_BATCH_BYTES {stages[0].PAIR_BATCH_BYTES >> 20} MiB): {base_ms:.3f} ms", flush=True)
        for bb, ns in ((None, 2), (None, 3), (None, 4), (1, 1), (1, 2), (1, 4)):
            for st in stages:
                st.PAIR_STREAMS = ns
                if bb is not None:
                    st.PAIR_BATCH_BYTES = bb
            ms, out = timed(call)
            same = torch.equal(out["depth"], base["depth"])
            print(f"   PAIR_STREAMS {ns}, {'one view per pass' if bb else 'default batching'}: {ms:.3f} ms, depth equal to default: {same}"
                  f" (max abs diff {float((out['depth'] - base['depth']).abs().max()):.2e})", flush=True)
        for st in stages:
            st.PAIR_STREAMS = 1
            st.PAIR_BATCH_BYTES = type(st).PAIR_BATCH_BYTES
    del net, scene
    torch.cuda.empty_cache()
