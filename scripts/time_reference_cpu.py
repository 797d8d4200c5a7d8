#!/usr/bin/env python3
"""Container-only cross-check of bench.py's cpu_baseline: times the REFERENCE's own hot path
(/root/reference models/MVSNet/model.py: build_cost_volume :109-139, cost_regularization :74-84, softmax + regression :207-209,
eval mode) and the oracle's streaming hot path on the same configuration-2 inputs and the same host threads.
Needs /root/reference; never runs on the GPU box.  Usage: python scripts/time_reference_cpu.py [--threads N] [--reps 3]"""
import argparse
import os
import statistics
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    import gen_golden
    gen_golden.import_reference()
    from models.MVSNet.model import MVSNet              # reference
    from models.MVSNet.module import depth_regression   # reference
    from wild_deep_mvs_amd import synthetic
    from oracle import mvsnet as O
    V, H, W, C, D = 5, 512, 640, 32, 192
    h, w = H // 4, W // 4
    net = MVSNet("variance")
    sd = synthetic.sharpened_state_dict("mvsnet", synthetic.template_of(net), seed=0)
    net.load_state_dict(sd)
    net.eval()
    net.num_depth = D
    cams = synthetic.make_cameras(1, V, H, W)
    Ks = cams["K"].clone(); Ks[:, :, :2] /= 4
    proj = O.build_proj_matrices(Ks, cams["R"], cams["t"])
    dv = cams["depth_min"][:, :1] + (cams["depth_max"][:, :1] - cams["depth_min"][:, :1]) / (D - 1) * torch.arange(D).view(1, -1)
    feats = synthetic.make_features(1, V, C, h, w, seed=1)
    fl = [feats[i] for i in range(V)]

    def ref_path():
        with torch.no_grad():
            cost = net.build_cost_volume(fl[0], fl[1:], proj[:, 0], [proj[:, i] for i in range(1, V)], dv)
            logits = net.cost_regularization(cost).squeeze(1)
            prob = torch.softmax(logits, dim=1)
            return depth_regression(prob, dv)

    def oracle_path():
        with torch.no_grad():
            return O.hot_path(fl, proj, dv.unsqueeze(1).expand(-1, V, -1), sd, streaming=True)[0]

    res = {}
    for name, fn in (("reference", ref_path), ("oracle_streaming", oracle_path)):
        fn()   # warm-up (allocator, thread pool)
        ts = []
        for _ in range(args.reps):
            t0 = time.perf_counter(); d = fn(); ts.append(time.perf_counter() - t0)
        res[name] = (statistics.median(ts), d)
        print(f"{name:18s} median of {args.reps}: {res[name][0]:.2f} s  ({D * h * w / res[name][0]:.3e} voxels/s) on {args.threads} threads")
    diff = float((res["reference"][1] - res["oracle_streaming"][1]).abs().max())
    print(f"max |depth_ref - depth_oracle| = {diff:.3e};  oracle / reference time = {res['oracle_streaming'][0] / res['reference'][0]:.2f}")


if __name__ == "__main__":
    main()
