#!/usr/bin/env python3
"""Dev (round 5): the fused tail (pscv_tail_sweep) against conv11^T + prob as two launches at the headline size
(u9 96x64x80x16, skip 192x128x160x8 -> logits 192x128x160), both storage formats; bit comparison; `tail_nbk` sweep."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wild_deep_mvs_amd import _lib as L, ops, synthetic  # noqa: E402
from wild_deep_mvs_amd.models.MVSNet.model import MVSNet  # noqa: E402


def timeit(fn, reps=30):
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
    net = MVSNet("variance")
    net.load_state_dict(synthetic.sharpened_state_dict("mvsnet", synthetic.template_of(net), seed=0))
    net = net.cuda().eval()
    for dt in (torch.float16, torch.bfloat16):
        ly = net.cost_regularization.engine_layers(dt)
        g = torch.Generator().manual_seed(0)
        u9 = (torch.randn(1, 96, 64, 80, 16, generator=g) * 0.5).to(dt).cuda()
        c0 = (torch.randn(1, 192, 128, 160, 8, generator=g) * 0.5).to(dt).cuda()
        two = lambda: ops.conv3d(ops.conv3d(u9, ly["conv11"], skip=c0), ly["prob"], out_dtype=torch.float32)
        ref = two().view(1, 192, 128, 160)
        print(f"{dt}: two launches {timeit(two):.1f} us", flush=True)
        for nbk in (0, 1, 2, 4, 8, 16, 32):
            L.set_tuning("tail_nbk", nbk)
            out = ops.tail_sweep(u9, ly["conv11"], ly["prob"], skip=c0)
            ne = int((out != ref).sum())
            us = timeit(lambda: ops.tail_sweep(u9, ly["conv11"], ly["prob"], skip=c0))
            print(f"   tail_sweep nbk={nbk:2d}: {us:6.1f} us, {ne} logits differ from the two launches", flush=True)
        L.set_tuning("tail_nbk", 0)


if __name__ == "__main__":
    main()
