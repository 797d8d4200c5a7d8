#!/usr/bin/env python3
"""Training-step timing of the MVSNet mirror on the engine: forward in train() mode + loss.backward() at
the headline size (5-view 512x640, D=192, B=1).  Prints one JSON line: ms per step, cost-volume voxels/s of a training
step and the per-kernel breakdown from HIP events on the launch stream.

    python scripts/bench_train.py [--steps 5] [--warmup 2] [--dtype bf16|f16] [--views 5] [--height 512] [--width 640] [--depth 192]
"""
import argparse
import gc
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wild_deep_mvs_amd import _lib as L  # noqa: E402
if os.environ.get("PSCV_LIB"):
    L.LIB_PATH = os.environ["PSCV_LIB"]      # A/B runs against another build of the library
from wild_deep_mvs_amd import ops, synthetic  # noqa: E402
from wild_deep_mvs_amd.models.MVSNet.model import MVSNet  # noqa: E402


def other_arch(a, dt):
    """Vis-MVSNet / CVP-MVSNet training step (train() forward on the engine's autograd nodes + the trainer's supervised loss +
    backward + Adam), same timing protocol as the MVSNet line."""
    if a.arch == "vis":
        from wild_deep_mvs_amd.models.VisMVSNet.frontend import Frontend
        net = Frontend()
        key, down, kw = "vis", 2, dict(depth_nums=[64, 32, 16], interval_scales=[2.0, 1.0, 0.5])
        net.depth_nums, net.interval_scales = kw["depth_nums"], kw["interval_scales"]
    else:
        from wild_deep_mvs_amd.models.CVP_MVSNet.frontend import Frontend
        net = Frontend()
        key, down, kw = "cvp", 1, dict(nscale=2)
    net.load_state_dict(synthetic.train_state_dict(key, synthetic.template_of(net), seed=0))
    net = net.cuda().train()
    net.train_storage_dtype = dt
    net.feature_engine_train = a.feature_engine          # the 2-D extractor / pyramid tower in train(): PyTorch-ROCm autograd or the engine's nodes
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    scene = synthetic.make_scene(a.batch, a.views, a.height, a.width, seed=0)
    if a.arch == "cvp":
        scene["t"] = scene["t"] * 8          # CVP's hypothesis spacing follows the baseline (as in the test fixtures)
    gt, mask = synthetic.train_target(scene, a.height // down, a.width // down)
    dev = {k: v.cuda() for k, v in scene.items() if isinstance(v, torch.Tensor)}
    gt, mask = gt.cuda(), mask.cuda()

    def step():
        opt.zero_grad(set_to_none=True)
        out = net(dev["imgs"], dev["K"], dev["R"], dev["t"], dev["depth_min"], dev["depth_max"], **kw)
        if a.arch == "vis":
            loss = synthetic.vis_supervised_loss(out, gt, mask, dev["depth_min"], dev["depth_max"], a.views)
        else:
            loss = synthetic.supervised_loss_list(out["depth_est_list"], gt, mask, dev["depth_min"], dev["depth_max"])
        loss.backward()
        opt.step()
        return loss

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    gc.disable()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / a.steps
    with ops.EventTimer() as tm:
        for _ in range(a.steps):
            step()
    gc.enable()
    kern = {k: round(v[1] * 1e3 / a.steps, 1) for k, v in sorted(tm.summary().items(), key=lambda kv: -kv[1][1])}
    print(json.dumps({"metric": f"{a.arch} training step (forward train() + backward + Adam)", "ms_per_step": ms, "dtype": a.dtype,
                      "loss": float(loss), "config": {"workload": f"{a.arch}, {a.views} views, {a.height}x{a.width}, B={a.batch}", **{k: str(v) for k, v in kw.items()}},
                      "pscv_kernels_us_per_step": dict(list(kern.items())[:12]), "pscv_kernels_total_ms": round(sum(kern.values()) / 1e3, 3)}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--views", type=int, default=5)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--depth", type=int, default=192)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--aggregation", default="variance")
    ap.add_argument("--arch", default="mvsnet", choices=["mvsnet", "vis", "cvp"],
                    help="vis: Vis-MVSNet (cascade depth_nums 64,32,16, the reference trainer's supervised loss incl. Bayesian pair terms); "
                         "cvp: CVP-MVSNet (nscale 2 as in training, supervised L1 on every level)")
    ap.add_argument("--no-pack-cache", action="store_true", help="rebuild every packed layer on every use (A/B of ops.PACK_CACHE)")
    ap.add_argument("--feature-engine", default="torch", choices=["torch", "pscv"],
                    help="2-D extractor in train(): PyTorch-ROCm autograd (fp32) or training.FeatureNetFn (engine, 16-bit activations)")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE", help="pscv_set_tuning knob (measurement runs)")
    a = ap.parse_args()
    for kv in a.tune:
        k, v = kv.split("=")
        L.set_tuning(k, int(v))
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[a.dtype]
    torch.cuda.set_device(0)
    ops.PACK_CACHE = not a.no_pack_cache
    if a.arch != "mvsnet":
        return other_arch(a, dt)
    net = MVSNet(a.aggregation)
    net.load_state_dict(synthetic.train_state_dict("mvsnet", synthetic.template_of(net), seed=0))
    net = net.cuda().train()
    net.num_depth = a.depth
    net.train_storage_dtype = dt
    net.feature_engine_train = a.feature_engine
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    scene = synthetic.make_scene(a.batch, a.views, a.height, a.width, seed=0)
    gt, mask = synthetic.train_target(scene, a.height // 4, a.width // 4)
    dev = {k: v.cuda() for k, v in scene.items() if isinstance(v, torch.Tensor)}
    gt, mask = gt.cuda(), mask.cuda()

    def step():
        opt.zero_grad(set_to_none=True)
        out = net(dev["imgs"], dev["K"], dev["R"], dev["t"], dev["depth_min"], dev["depth_max"])
        loss = synthetic.supervised_loss(out["depth"], gt, mask, dev["depth_min"], dev["depth_max"])
        loss.backward()
        opt.step()
        return loss

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    gc.disable()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / a.steps
    with ops.EventTimer() as tm:
        for _ in range(a.steps):
            step()
    summ = tm.summary()
    gc.enable()
    kern = {k: round(v[1] * 1e3 / a.steps, 1) for k, v in sorted(summ.items(), key=lambda kv: -kv[1][1])}
    vox = a.batch * a.depth * (a.height // 4) * (a.width // 4)
    print(json.dumps({"metric": "MVSNet training step (forward train() + backward + Adam), cost-volume voxels/s", "value": vox / (ms * 1e-3),
                      "unit": "voxels/s", "ms_per_step": ms, "dtype": a.dtype, "loss": float(loss),
                      "config": {"workload": f"MVSNet {a.aggregation}, {a.views} views, {a.height}x{a.width}, D={a.depth}, B={a.batch}",
                                 "feature_engine_train": a.feature_engine},
                      "pscv_kernels_us_per_step": kern, "pscv_kernels_total_ms": round(sum(kern.values()) / 1e3, 3)}))


if __name__ == "__main__":
    main()
