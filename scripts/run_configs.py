#!/usr/bin/env python3
"""Run the five BASELINE.json configurations (single-GPU forms) through the engine: full forward() incl. the 2-D feature
nets, synthetic inputs, sharpened weights.  Prints time per forward, cost-volume voxels and sanity of the outputs.
Usage: python scripts/run_configs.py [--only N] [--reps 3]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wild_deep_mvs_amd import synthetic  # noqa: E402


def build(arch):
    if arch in ("mvsnet", "mvsnet_s"):
        from wild_deep_mvs_amd.models.MVSNet.model import MVSNet
        net = MVSNet("variance" if arch == "mvsnet" else "softmin")
        key = "mvsnet"
    elif arch == "vis":
        from wild_deep_mvs_amd.models.VisMVSNet.frontend import Frontend
        net, key = Frontend(), "vis"
    else:
        from wild_deep_mvs_amd.models.CVP_MVSNet.frontend import Frontend
        net, key = Frontend(), "cvp"
    net.load_state_dict(synthetic.sharpened_state_dict(key, synthetic.template_of(net), seed=0))
    return net.cuda().eval()


CONFIGS = {
    1: dict(arch="mvsnet_s", V=3, H=128, W=160, setup=lambda n: setattr(n, "num_depth", 48), kw={}, vox=lambda: 48 * 32 * 40),
    2: dict(arch="mvsnet", V=5, H=512, W=640, setup=lambda n: None, kw={}, vox=lambda: 192 * 128 * 160),
    3: dict(arch="vis", V=5, H=512, W=640, setup=lambda n: None,
            kw=dict(depth_nums=[192, 32, 16], interval_scales=[128 / 192, 1, 0.5]),
            vox=lambda: 192 * 64 * 80 + 32 * 128 * 160 + 16 * 256 * 320),
    4: dict(arch="cvp", V=5, H=1024, W=1280, setup=lambda n: None, kw=dict(nscale=5),
            vox=lambda: 96 * 64 * 80 + 8 * (128 * 160 + 256 * 320 + 512 * 640 + 1024 * 1280), bscale=8),
    5: dict(arch="vis", V=9, H=1152, W=1600, setup=lambda n: None, kw=dict(depth_nums=[256, 32, 16], interval_scales=[0.5, 1, 0.5]),
            vox=lambda: 256 * 144 * 200 + 32 * 288 * 400 + 16 * 576 * 800),
}


HBM_PEAK_GBS, MFMA_PEAK_TFLOPS = 8000.0, 2500.0     # /opt/skills/guides/MI355X_MICROARCH.md


def kernel_rooflines(detail: dict, top: int = 3) -> list:
    """The ``top`` kernels by GPU time of one eager forward (ops.EventTimer.detail()): algorithmic bytes / flops per launch,
    average duration, and the fraction of the HBM and dense-MFMA peaks they reach; ``bound`` = the roofline that is nearer
    (arithmetic intensity against the 312 FLOP/B ridge)."""
    total = sum(v["ms"] for v in detail.values()) or 1.0
    rows = []
    for name, v in sorted(detail.items(), key=lambda kv: -kv[1]["ms"])[:top]:
        t = v["ms"] * 1e-3
        gbs = v["bytes"] / t / 1e9 if v["bytes"] else None
        tfs = v["flops"] / t / 1e12 if v["flops"] else None
        mfma = name.startswith("conv")
        ai = v["flops"] / v["bytes"] if v["bytes"] else None
        bound = "mfma" if (mfma and ai is not None and ai > MFMA_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS) else "hbm"
        rows.append({"kernel": name, "launches": v["launches"], "avg_us": v["ms"] / v["launches"] * 1e3, "share_of_gpu_time": v["ms"] / total,
                     "bound": bound, "achieved_GBs": gbs, "hbm_frac": None if gbs is None else gbs / HBM_PEAK_GBS,
                     "achieved_TFLOPs": tfs if mfma else None, "mfma_frac": (tfs / MFMA_PEAK_TFLOPS) if (mfma and tfs) else None,
                     "frac": (tfs / MFMA_PEAK_TFLOPS) if bound == "mfma" else (None if gbs is None else gbs / HBM_PEAK_GBS)})
    return rows


def time_config(cid: int, reps: int = 3) -> dict:
    """One BASELINE configuration through the engine's full forward() (2-D features included), driven the way the reference's
    callers drive it -- ``net(...)`` on the same signature again and again (depthmap_eval.py:106): ms per forward with the
    in-forward hipGraph replay (the default) and with eager launches (``graph_replay = False``), wall clock between
    synchronisations after warm-up; then one eager pass under HIP events for the per-kernel rooflines.  Used by bench.py's
    "other_configs" and by this script."""
    import gc
    from wild_deep_mvs_amd import ops
    cfg = CONFIGS[cid]
    net = build(cfg["arch"])
    cfg["setup"](net)
    scene = synthetic.make_scene(1, cfg["V"], cfg["H"], cfg["W"], seed=cid)
    if "bscale" in cfg:
        scene["t"] = scene["t"] * cfg["bscale"]
    dev = {k: v.cuda() for k, v in scene.items()}
    call = lambda: net(dev["imgs"], dev["K"], dev["R"], dev["t"], dev["depth_min"], dev["depth_max"], **cfg["kw"])

    def timed():
        out = call(); out = call(); out = call()        # (eager, capture, replay -- or three eager calls)
        torch.cuda.synchronize()
        gc.collect(); gc.disable()
        try:
            ts = []
            for _ in range(max(reps, 5)):                # median of individually timed calls: one allocator / driver hiccup
                t0 = time.perf_counter()                 # (a 50 ms outlier was seen once) must not become the number
                out = call()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            return sorted(ts)[len(ts) // 2], out
        finally:
            gc.enable()
    with torch.no_grad():
        net.graph_replay = False
        dt_eager, out = timed()
        detail = None
        for _ in range(3):                               # three event passes, the one with the least GPU time is reported: a single
            with ops.EventTimer() as tm:                 # pass once showed five launches of one layer at 10x their usual duration
                call()                                   # (a shared box), which put that layer on top of a configuration's table
            dtl = tm.detail()
            if detail is None or sum(v["ms"] for v in dtl.values()) < sum(v["ms"] for v in detail.values()):
                detail = dtl
        net.graph_replay = True
        dt, gout = timed()
        same = float((gout["depth"] - out["depth"]).abs().max())
    d = out["depth"]
    gpu_ms = sum(v["ms"] for v in detail.values())
    res = {"config": cid, "model": cfg["arch"], "views": cfg["V"], "image": [cfg["H"], cfg["W"]], "kwargs": {k: v for k, v in cfg["kw"].items()},
           "ms_per_forward": dt * 1e3, "voxels": cfg["vox"](), "voxels_per_s": cfg["vox"]() / dt,
           "finite": bool(torch.isfinite(d).all()) and bool(torch.isfinite(out["photometric_confidence"]).all()),
           "timed": "full forward() incl. 2-D feature nets called like depthmap_eval.py:106 (the forward replays its own hipGraph from "
                    "the second call of a signature on), fp16 storage",
           "ms_per_forward_eager": dt_eager * 1e3, "replay_max_abs_diff_vs_eager": same,
           "engine_kernel_ms_eager_pass": gpu_ms, "engine_launches": sum(v["launches"] for v in detail.values()),
           "roofline": kernel_rooflines(detail)}
    if cid == 4:
        # BASELINE.json names "per-level D = 48" for this configuration: the reference's TRAIN-mode plane count of the coarsest level
        # (net.py:126-127); its eval mode -- what the line above times -- sweeps 96.  The same forward with 48 coarse planes:
        with torch.no_grad():
            net.model.coarse_planes_eval = 48
            dt48, out48 = timed()
            net.model.coarse_planes_eval = 96
        vox48 = cfg["vox"]() - 48 * 64 * 80
        res["coarse_48_planes"] = {"ms_per_forward": dt48 * 1e3, "voxels": vox48, "voxels_per_s": vox48 / dt48,
                                   "finite": bool(torch.isfinite(out48["depth"]).all()),
                                   "what": "the reference's train-mode coarse plane count (BASELINE.json's 'per-level D=48') in the eval forward"}
        del out48
    del net, out, gout, dev
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", type=int, default=0)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--feature-engine", default="", help="pscv | torch: 2-D extractor (default: the model's)")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE", help="pscv_set_tuning knob (measurement runs)")
    ap.add_argument("--graph", action="store_true", help="also time hipGraph replay of the forward (wild_deep_mvs_amd.graph.GraphedModel)")
    args = ap.parse_args()
    for kv in args.tune:
        from wild_deep_mvs_amd import _lib
        k, v = kv.split("=")
        _lib.set_tuning(k, int(v))
    for cid, cfg in CONFIGS.items():
        if args.only and cid != args.only:
            continue
        net = build(cfg["arch"])
        cfg["setup"](net)
        if args.feature_engine and hasattr(net, "feature_engine"):
            net.feature_engine = args.feature_engine
        scene = synthetic.make_scene(1, cfg["V"], cfg["H"], cfg["W"], seed=cid)
        if "bscale" in cfg:
            scene["t"] = scene["t"] * cfg["bscale"]
        dev = {k: v.cuda() for k, v in scene.items()}
        call = lambda: net(dev["imgs"], dev["K"], dev["R"], dev["t"], dev["depth_min"], dev["depth_max"], **cfg["kw"])
        import gc
        out = call()
        out = call()
        torch.cuda.synchronize()
        gc.collect(); gc.disable()      # a generation-2 collection inside the loop costs ~40 ms (seen in bench.py too)
        t0 = time.perf_counter()
        for _ in range(args.reps):
            out = call()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.reps
        gc.enable()
        dt_graph = None
        if args.graph:
            from wild_deep_mvs_amd.graph import GraphedModel
            try:
                gnet = GraphedModel(net)
                gcall = lambda: gnet(dev["imgs"], dev["K"], dev["R"], dev["t"], dev["depth_min"], dev["depth_max"], **cfg["kw"])
                gcall(); gcall()
                torch.cuda.synchronize()
                gc.disable()
                t0 = time.perf_counter()
                for _ in range(args.reps):
                    gcall()
                torch.cuda.synchronize()
                dt_graph = (time.perf_counter() - t0) / args.reps
                gc.enable()
                del gnet
            except Exception as e:   # e.g. MVSNet-s reads its temperature on the host (a sync inside the capture)
                print(f"   (config {cid}: hipGraph capture not possible: {type(e).__name__}: {str(e)[:120]})")
                gc.enable()
        d = out["depth"]
        ok = bool(torch.isfinite(d).all()) and bool(torch.isfinite(out["photometric_confidence"]).all())
        print(f"config {cid} {cfg['arch']:8s} V={cfg['V']} {cfg['H']}x{cfg['W']}: {dt * 1e3:8.2f} ms / forward (with 2-D features), "
              f"{cfg['vox']() / dt / 1e9:6.3f} G cost-volume voxels/s, depth {tuple(d.shape)} range {float(d.min()):.3f}..{float(d.max()):.3f}, "
              f"finite={ok}, peak mem {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB"
              + (f", hipGraph replay {dt_graph * 1e3:.2f} ms" if dt_graph is not None else ""), flush=True)
        del net, out
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()


if __name__ == "__main__":
    main()
