#!/usr/bin/env python3
"""Vis-MVSNet's 2-D extractor in train() stays on PyTorch-ROCm: does its memory format / MIOpen's find mode matter?
Times feat_ext forward + backward for 5 views of 512x640 (one call per view, like the frontend) in NCHW and channels_last,
with and without torch.backends.cudnn.benchmark."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wild_deep_mvs_amd.models.VisMVSNet.model_cas import FeatExt

def run(cl, bench):
    torch.backends.cudnn.benchmark = bench
    net = FeatExt().cuda().train()
    if cl:
        net = net.to(memory_format=torch.channels_last)
    imgs = [torch.rand(1, 3, 512, 640, device="cuda") for _ in range(5)]
    if cl:
        imgs = [i.contiguous(memory_format=torch.channels_last) for i in imgs]
    def step():
        net.zero_grad(set_to_none=True)
        loss = 0
        for im in imgs:
            o = net(im)
            loss = loss + sum(x.float().square().mean() for x in o)
        loss.backward()
    for _ in range(4): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 5 * 1e3

for cl in (False, True):
    for bench in (False, True):
        print(f"channels_last={cl} cudnn.benchmark={bench}: {run(cl, bench):.2f} ms per step (5 views fwd + bwd)", flush=True)
