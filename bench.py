#!/usr/bin/env python3
"""Headline benchmark: cost-volume voxels/s of the MVSNet depth-inference hot path on MI355X.

A step = one pass of the hot path (fused warp + variance cost volume -> 3-D U-Net regularisation ->
softmax / depth regression / confidence) over one batch of synthetic input that is already resident in
HBM: BASELINE.json configs[1] = MVSNet, 1 ref + 4 src views, 512x640 images (128x160x32 feature maps),
D = 192 planes, bf16 storage / fp32 accumulation (--dtype bf16, the default since round 6; fp16 is measured too and reported under "alt").
3 932 160 cost-volume voxels per reference view; a batch is --batch reference views (default 3, each with its own source views).
--batch-mode views (default): one single-branch hipGraph per view, each replayed on its own HIP stream, consecutive steps not joined
(wild_deep_mvs_amd.graph.ViewPipeline); --batch-mode streams (rounds 3-5): ONE hipGraph whose views are parallel branches;
--batch-mode batched: one launch per layer for the whole batch on one stream, replayed as a hipGraph; --batch 1 = one view at a time as in
rounds 1-2, also reported in every line.  The timed region (exactly --steps steps between barrier + synchronize) runs --repeats times
(default 5); the line reports the median region and min / median / max.

Multi-GPU (--gpus N, one process per GPU): reference views are independent objects, so each rank sweeps its own
view (global batch = N) with no data-path collective -> weak scaling.  Ranks come either from a launcher
(`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`: RANK / WORLD_SIZE / MASTER_* in the
environment) or, when WORLD_SIZE is unset, from bench.py itself (`python bench.py --gpus N` spawns its N ranks the
way the reference's train.py / depthmap_eval.py do with mp.spawn).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     the dominant kernel's achieved algorithmic GB/s vs the HBM peak (HIP events on the launch stream)
  cpu_baseline the CPU oracle (a port of the reference's PyTorch path) timed on this box's host cores
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from wild_deep_mvs_amd import ops, synthetic  # noqa: E402
from wild_deep_mvs_amd.models.MVSNet.model import MVSNet, build_proj_matrices  # noqa: E402

V, IMG_H, IMG_W, C, D = 5, 512, 640, 32, 192
h, w = IMG_H // 4, IMG_W // 4
VOX = D * h * w
HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes(name: str) -> float:
    """Algorithmic HBM bytes of one launch at the headline size."""
    if name.startswith("warp_cost"):
        return V * C * h * w * 2 + C * VOX * 2                      # read V feature maps + write the volume once
    if name.startswith("conv3d[32->8,k"):
        return C * VOX * 2 + 8 * VOX * 2                            # read 32-ch volume, write 8-ch volume
    if name.startswith("conv3d[16->8,k"):
        return 16 * (VOX // 8) * 2 + 2 * 8 * VOX * 2                # read half-res 16ch + skip, write 8ch
    if name.startswith("conv3d[8->1,k"):
        return 8 * VOX * 2 + VOX * 4
    if name.startswith("conv3d[8->16,k1]"):
        return 8 * VOX * 2 + 16 * (VOX // 8) * 2
    if name.startswith("tail_sweep"):
        return 16 * (VOX // 8) * 2 + 8 * VOX * 2 + VOX * 4          # read half-res 16ch + the 8-ch skip, write fp32 logits (no intermediate)
    if name.startswith("softargmin"):
        return VOX * 4 + 2 * h * w * 4
    return 0.0


MFMA_PEAK_TFLOPS = 2500.0   # dense bf16 / fp16 MFMA peak of one MI355X (MI355X_MICROARCH.md)


DTYPES = {"f16": torch.float16, "bf16": torch.bfloat16}


def pmc_traffic(name: str):
    """HBM bytes per launch of `name` from the newest committed rocprofv3 PMC capture (profiles/rNN_traffic.json, made by
    scripts/profile.sh + scripts/prof_summary.py on the same command: FETCH_SIZE doubled per the gfx950 note, WRITE_SIZE as is;
    both 16-bit formats launch the same grid and move the same bytes), with the file it came from -- or (None, None).  The
    capture is a separate rocprofv3 run (counters cannot be read from inside the timed process): regenerate it whenever the
    dominant kernel changes; the file name in the line says which capture the number is from."""
    import glob
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]_traffic.json")), reverse=True):
        try:
            table = json.load(open(path))
        except Exception:
            continue
        for key, rec in table.items():
            if name.startswith(key) and rec.get("hbm_bytes"):
                return rec["hbm_bytes"], os.path.basename(path)
    return None, None


def live_traffic(dtype_name: str, timeout_s: int = 150):
    """HBM bytes per launch of the headline step's full-resolution kernels, measured NOW: two short rocprofv3 counter passes
    (FETCH_SIZE and WRITE_SIZE each in its own run, kernel trace only -- MI355X_MICROARCH.md's HBM recipe) over this same script
    with `--steps 3 --eager`, read back from the rocpd databases by scripts/prof_summary.py.  Returns ({kernel: bytes}, note) or
    (None, reason).  Runs after the timed regions, in child processes; the committed capture stays the fallback."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    root = tempfile.mkdtemp(prefix="pscv_traffic_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs",
           "--eager", "--no-live-traffic", "--batch", "1", "--dtype", dtype_name]
    try:
        for sub, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
            r = subprocess.run([exe, "--kernel-trace", "--pmc", counter, "-d", os.path.join(root, sub), "-o", "bench", "--"] + cmd,
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} rc={r.returncode}: {r.stderr[-200:]}"
        sys.path.insert(0, os.path.join(REPO, "scripts"))
        import prof_summary
        prof_summary.traffic_json(root)
        table = json.load(open(os.path.join(root, "traffic.json")))
        return {k: v["hbm_bytes"] for k, v in table.items()}, "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run (FETCH_SIZE x 2: gfx950)"
    except Exception as e:   # pragma: no cover
        return None, f"{type(e).__name__}: {e}"[:200]
    finally:
        shutil.rmtree(root, ignore_errors=True)


def build_inputs(device, rank: int, dtype, batch: int = 1):
    net = MVSNet("variance")
    sd = synthetic.sharpened_state_dict("mvsnet", synthetic.template_of(net), seed=0)
    net.load_state_dict(sd)
    net = net.to(device).eval()
    net.num_depth = D
    net.storage_dtype = dtype
    cams = synthetic.make_cameras(batch, V, IMG_H, IMG_W)
    Ks = cams["K"].clone()
    Ks[:, :, :2] /= 4
    proj = build_proj_matrices(Ks, cams["R"], cams["t"])
    steps = torch.arange(D, dtype=torch.float32).view(1, -1)
    dv = cams["depth_min"][:, :1] + (cams["depth_max"][:, :1] - cams["depth_min"][:, :1]) / (D - 1) * steps
    feats = synthetic.make_features(batch, V, C, h, w, seed=1 + rank)           # [V,batch,C,h,w] fp32: every reference view its own maps
    feats_cl = [ops.to_channels_last(feats[i].to(device), dtype) for i in range(V)]
    return net, sd, feats, feats_cl, proj.to(device), dv.to(device).contiguous(), proj, dv


def rig_step(net, device, dtype, rig: str, nb: int, steps: int = 30, views: bool = True):
    """The headline step (a batch of `nb` reference views, hot path; `views`: free-running per-view hipGraphs like the headline region,
    else one replayed hipGraph of `net.hot_path` on the batch) on the camera rig `rig`: ms per step."""
    cams = synthetic.make_cameras(nb, V, IMG_H, IMG_W, rig=rig)
    Ks = cams["K"].clone()
    Ks[:, :, :2] /= 4
    proj = build_proj_matrices(Ks, cams["R"], cams["t"]).to(device)
    st = torch.arange(D, dtype=torch.float32).view(1, -1)
    dv = (cams["depth_min"][:, :1] + (cams["depth_max"][:, :1] - cams["depth_min"][:, :1]) / (D - 1) * st).to(device).contiguous()
    feats = synthetic.make_features(nb, V, C, h, w, seed=11)
    fcl = [ops.to_channels_last(feats[i].to(device), dtype) for i in range(V)]
    with torch.no_grad():
        for _ in range(3):
            net.hot_path(fcl, proj, dv)
        torch.cuda.synchronize()
        if views and nb >= 2:          # the headline's step: free-running per-view graphs
            from wild_deep_mvs_amd.graph import ViewPipeline
            pipe = ViewPipeline(net, fcl, proj, dv)
            run, finish = pipe.step, pipe.results
        else:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                net.hot_path(fcl, proj, dv)
            run, finish = g.replay, (lambda: None)
        for _ in range(10):
            run()
        finish()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        finish()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3


def staging_modes(device, dtype, rig: str):
    """Staging-mode histogram of ONE LDS-staged warp launch at the headline size on camera rig `rig`: (shares over all (workgroup,
    plane range, source view) triples, per-view counts).  Modes: see ``geometry_probe``."""
    import ctypes
    from wild_deep_mvs_amd import _lib
    cams = synthetic.make_cameras(1, V, IMG_H, IMG_W, rig=rig)
    Ks = cams["K"].clone()
    Ks[:, :, :2] /= 4
    proj = build_proj_matrices(Ks, cams["R"], cams["t"]).to(device)
    steps = torch.arange(D, dtype=torch.float32).view(1, -1)
    dv = (cams["depth_min"][:, :1] + (cams["depth_max"][:, :1] - cams["depth_min"][:, :1]) / (D - 1) * steps).to(device).contiguous()
    feats = synthetic.make_features(1, V, C, h, w, seed=7)
    fcl = [ops.to_channels_last(feats[i].to(device), dtype) for i in range(V)]
    cam_blocks = ops.proj_cams_device(proj.float().contiguous(), 0)
    hist = torch.zeros(16, dtype=torch.int32, device=device)
    fn = _lib.lib().pscv_debug_wl_mode_hist
    fn.argtypes, fn.restype = [ctypes.c_void_p], None
    fn(hist.data_ptr())
    try:
        ops.warp_cost(fcl[0], fcl[1:], cam_blocks, dv, cost=_lib.COST_VARIANCE, out_dtype=dtype)
        torch.cuda.synchronize()
    finally:
        fn(None)
    hm = hist.view(4, 4).cpu().tolist()
    total = max(1, sum(hm[0]))
    names = ("DIRECT", "GEN", "FAST", "ZERO")
    return ({names[m]: round(sum(hm[v][m] for v in range(V - 1)) / (total * (V - 1)), 4) for m in range(4)},
            [dict(zip(names, hm[v])) for v in range(V - 1)])


def geometry_probe(net, device, dtype, reps: int = 20, nb: int = 1, views: bool = True):
    """The headline workload on BOTH camera rigs of `synthetic.make_cameras`: "probe" (what the bench line is
    measured on: depth 2..6, sources rotated about y and shifted along x, 0.03-0.1 feature texels per plane) and "dtu" (depth
    425..905 as data/dtu_yao.py:109, cameras on an arc with tilt, 0.14-0.32 texels per plane, oblique epipolar lines).  Per rig: the
    warp + cost launch alone (HIP events; the three warp kernels interleaved, medians of five rounds), one eager hot path, and the LDS-staged kernel's staging-mode histogram
    -- per (workgroup, source view): FAST = box staged, no masks; GEN = staged, clipped at the image border; DIRECT = box too
    large for the LDS budget or a corner behind the camera: global taps; ZERO = box outside the image, the view contributes 0."""
    import ctypes
    from wild_deep_mvs_amd import _lib
    out = {}
    for rig in ("probe", "dtu"):
        cams = synthetic.make_cameras(1, V, IMG_H, IMG_W, rig=rig)
        Ks = cams["K"].clone()
        Ks[:, :, :2] /= 4
        proj = build_proj_matrices(Ks, cams["R"], cams["t"]).to(device)
        steps = torch.arange(D, dtype=torch.float32).view(1, -1)
        dv = (cams["depth_min"][:, :1] + (cams["depth_max"][:, :1] - cams["depth_min"][:, :1]) / (D - 1) * steps).to(device).contiguous()
        feats = synthetic.make_features(1, V, C, h, w, seed=7)
        fcl = [ops.to_channels_last(feats[i].to(device), dtype) for i in range(V)]
        cam_blocks = ops.proj_cams_device(proj.float().contiguous(), 0)
        warp = lambda: ops.warp_cost(fcl[0], fcl[1:], cam_blocks, dv, cost=_lib.COST_VARIANCE, out_dtype=dtype)
        hist = torch.zeros(16, dtype=torch.int32, device=device)
        fn = _lib.lib().pscv_debug_wl_mode_hist
        fn.argtypes, fn.restype = [ctypes.c_void_p], None
        fn(hist.data_ptr())
        try:
            warp()
            torch.cuda.synchronize()
        finally:
            fn(None)
        hm = hist.view(4, 4).cpu().tolist()
        total = max(1, sum(hm[0]))
        names = ("DIRECT", "GEN", "FAST", "ZERO")
        modes = {names[m]: round(sum(hm[v][m] for v in range(V - 1)) / (total * (V - 1)), 4) for m in range(4)}
        # the three kernels INTERLEAVED over several rounds (round 0 = warm-up), medians: stand-alone launches one kernel after the
        # other measured the clock ramp of whichever came first (round 4's line: 213.7 us for a kernel that an interleaved run puts at 176)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        kernels = (("lds_staged", -1), ("direct_gather", 0), ("lane_owner", 4))   # -1: default; 0: geometry-independent global taps; 4: csrc/warp_cost_lv.hip
        samples = {k: [] for k, _ in kernels}
        try:
            for rnd in range(6):
                for key, tiled in kernels:
                    _lib.set_tuning("warp_tiled", tiled)
                    for _ in range(2):
                        warp()
                    e0.record()
                    for _ in range(reps):
                        warp()
                    e1.record()
                    torch.cuda.synchronize()
                    if rnd:
                        samples[key].append(e0.elapsed_time(e1) * 1e3 / reps)
        finally:
            _lib.set_tuning("warp_tiled", -1)
        med = lambda v: sorted(v)[len(v) // 2]
        warp_us = med(samples["lds_staged"])
        alt_us = {"direct_gather": med(samples["direct_gather"]), "lane_owner": med(samples["lane_owner"])}
        with torch.no_grad():
            for _ in range(2):
                net.hot_path(fcl, proj, dv)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                net.hot_path(fcl, proj, dv)
            torch.cuda.synchronize()
            path_ms = (time.perf_counter() - t0) / reps * 1e3
        quad_us = alt_us["direct_gather"]
        try:
            step_ms = rig_step(net, device, dtype, rig, nb, views=views)
        except Exception as e:   # pragma: no cover
            step_ms = None
        out[rig] = {"step_ms": None if step_ms is None else round(step_ms, 4), "views_per_step": nb,
                    "value": None if step_ms is None else nb * VOX / (step_ms * 1e-3), "unit": "voxels/s",
                    "warp_cost_us": round(warp_us, 1), "warp_cost_hbm_frac": round(algorithmic_bytes("warp_cost[0]") / (warp_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                    "warp_cost_us_direct_gather_kernel": round(quad_us, 1), "warp_cost_us_lane_owner_kernel": round(alt_us["lane_owner"], 1),
                    "hot_path_eager_ms_per_view": round(path_ms, 4), "staging_modes_share_of_block_views": modes,
                    "staging_modes_per_view": [dict(zip(names, hm[v])) for v in range(V - 1)]}
    out["note"] = ("`value` / `step_ms`: the headline step (same batch, same launch scheme as the timed region) on each rig with DEFAULT tuning -- the top-level "
                   "`value` is the probe rig's, `value_dtu_rig` repeats the DTU-like rig's.  On the DTU-like rig the source boxes of a whole chunk (48 planes since round 6) "
                   "exceed the LDS-staged kernel's 16 x 8 texel / arena budget for the wide-baseline views; such a block sweeps its chunk as two "
                   "halves with their own boxes (round 5), and a half in which some view would still take global taps (DIRECT) as two quarters "
                   "(round 6; seven boxes per view from ONE box phase): the histogram counts (swept plane range, view) pairs.  `warp_cost_us*`: "
                   "stand-alone launches of the three warp kernels, interleaved, medians of five rounds (scripts/dev/warp_ab.py adds the split levels: "
                   "quarters 146-148 us, halves only 153-173 us, no split 180-206 us, direct-gather kernel 149-150 us on the DTU-like rig; "
                   "117-119 / 118 / 123-125 / 146 us on the probe rig; profiles/r06_warp_quarter_split.txt).  The DTU-like rig has 97 % of its (block, view) pairs inside "
                   "the source images against 80 % on the probe rig (ZERO 0.03 / 0.18): more real work per voxel, not only larger boxes")
    return out


REF_CONTAINER = {"seconds": 3.47, "voxels_per_s": VOX / 3.47, "threads": 8,
                 "what": "the reference's own eval path (/root/reference models/MVSNet/model.py:109-139,74-84,207-209) on the same "
                         "inputs in the build container (8 vCPU), median of 3; the oracle's streaming path took 2.95 s there with "
                         "identical depth (scripts/time_reference_cpu.py)"}


def physical_cores() -> int:
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def cpu_baseline(sd, feats, proj, dv, gpu_depths: dict):
    """The oracle's hot path in the reference's eval-mode order (view by view, in-place sums: oracle.mvsnet.hot_path(streaming=True),
    same stages, fp32 ATen) on ONE reference view of the step's batch (a bounded sample of the same workload): one warm-up pass, then
    the median of three, on the physical host cores.  One more pass per remaining batch item (untimed) so that the second half of
    BASELINE.json's metric -- relative L1 of the engine's depth maps (one per storage format) against the oracle at the full
    headline size -- covers every view of the step."""
    from oracle import mvsnet as O          # cpu_baseline leg only
    cores = physical_cores()
    torch.set_num_threads(cores)
    B = feats.shape[1]
    dvv = dv.unsqueeze(1).expand(-1, V, -1)
    item = lambda b: ([feats[i][b:b + 1] for i in range(V)], proj[b:b + 1], dvv[b:b + 1])
    times, o_depths = [], []
    with torch.no_grad():
        fl, pj, dd = item(0)
        o_depth, _ = O.hot_path(fl, pj, dd, sd, streaming=True)     # warm-up: thread pool, allocator
        for _ in range(3):
            t0 = time.perf_counter()
            o_depth, _ = O.hot_path(fl, pj, dd, sd, streaming=True)
            times.append(time.perf_counter() - t0)
        o_depths.append(o_depth)
        for b in range(1, B):
            fl, pj, dd = item(b)
            o_depths.append(O.hot_path(fl, pj, dd, sd, streaming=True)[0])
    o_all = torch.cat(o_depths, 0)
    dt = sorted(times)[1]
    rel = {k: float((d.float().cpu() - o_all).abs().mean() / o_all.abs().mean()) for k, d in gpu_depths.items()}
    return {"value": VOX / dt, "unit": "voxels/s", "cores": cores, "kind": "port",
            "sample": f"one reference view of the step's batch (5-view 128x160x32 features, D=192, fp32), eval-mode order of the reference; "
                      f"warm-up + median of 3 passes ({dt:.2f} s; all: {', '.join(f'{t:.2f}' for t in times)}); the depth comparison covers all "
                      f"{B} views of the step",
            "ref_container_s": REF_CONTAINER["seconds"], "ref_container": REF_CONTAINER}, rel


SHARDED = {
    # BASELINE configuration 2 (the headline workload, full forward): MVSNet 5-view 512x640 D=192 with the depth planes of the WHOLE
    # hot path sharded (MVSNet.set_depth_group: per-layer one-plane halo exchange, LSE-merged regression) -- and the size SURVEY 8e
    # names for where that shard can pay: 1152x1600 images (288x400 maps), D = 256, 29 M cost-volume voxels
    "mvsnet_depth": dict(config=2, model="mvsnet", V=5, H=512, W=640, num_depth=192, kw={}, vox=192 * 128 * 160, setter="set_depth_group"),
    "mvsnet_depth_large": dict(config="2-large", model="mvsnet", V=5, H=1152, W=1600, num_depth=256, kw={}, vox=256 * 288 * 400,
                               setter="set_depth_group"),
    # BASELINE configuration 3: Vis-MVSNet 5-view 512x640, depth planes [192,32,16] sharded over the ranks (LSE-merged heads)
    "depth": dict(config=3, model="vis", V=5, H=512, W=640, kw=dict(depth_nums=[192, 32, 16], interval_scales=[128 / 192, 1, 0.5]),
                  vox=192 * 64 * 80 + 32 * 128 * 160 + 16 * 256 * 320, setter="set_depth_group"),
    # the same configuration with NO replicated stage (round 4): stage 1 by depth planes, the per-pixel stages 2-3 by image rows
    # (SingleStage.forward_row_shard: slab + recomputed 16-row halo, one all-gather of the small output maps per stage)
    "depth_rows": dict(config=3, model="vis", V=5, H=512, W=640, kw=dict(depth_nums=[192, 32, 16], interval_scales=[128 / 192, 1, 0.5]),
                       vox=192 * 64 * 80 + 32 * 128 * 160 + 16 * 256 * 320, setter="set_depth_row_groups"),
    # BASELINE configuration 5: Vis-MVSNet 9-view 1152x1600 [256,32,16], the 8 source views sharded over the ranks (16-bit shares of
    # the visibility-weighted sums reduce-scattered into depth / row slabs, slab-sharded RegFuse: the "RCCL variance reduce" of that model)
    # BASELINE configuration 4: CVP-MVSNet 5-view 1024x1280, five pyramid levels; the image rows of the four refinement levels (8
    # per-pixel planes each, 10.5 M voxels at the finest) sharded over the ranks (round 5: slab + 20-row recomputed halo, one all-gather
    # of the level's depth rows; the coarsest level and the pyramid tower run replicated)
    "cvp_rows": dict(config=4, model="cvp", V=5, H=1024, W=1280, kw=dict(nscale=5), vox=14417920, setter="set_row_group"),
    "view": dict(config=5, model="vis", V=9, H=1152, W=1600, kw=dict(depth_nums=[256, 32, 16], interval_scales=[0.5, 1, 0.5]),
                 vox=256 * 144 * 200 + 32 * 288 * 400 + 16 * 576 * 800, setter="set_view_group"),
}


def _sharded_net(cfg, device):
    if cfg["model"] == "mvsnet":
        net = MVSNet("variance")
        net.load_state_dict(synthetic.sharpened_state_dict("mvsnet", synthetic.template_of(net), seed=0))
        net.num_depth = cfg["num_depth"]
    elif cfg["model"] == "cvp":
        from wild_deep_mvs_amd.models.CVP_MVSNet.frontend import Frontend as CvpFrontend
        net = CvpFrontend()
        net.load_state_dict(synthetic.sharpened_state_dict("cvp", synthetic.template_of(net), seed=0))
    else:
        from wild_deep_mvs_amd.models.VisMVSNet.frontend import Frontend
        net = Frontend()
        net.load_state_dict(synthetic.sharpened_state_dict("vis", synthetic.template_of(net), seed=0))
    return net.to(device).eval()


def _all_ok(dist, device, ok: bool) -> bool:
    """True iff every rank reports ``ok`` (one MIN all-reduce of a flag): ranks agree to skip a leg together instead of one of them
    raising while the others wait inside the next collective."""
    flag = torch.tensor([1 if ok else 0], device=device, dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())


def sharded_legs(dist, device, world, rank, reps=5, only=None):
    """N > 1 only, after the headline region: ONE reference view computed cooperatively by all ranks (strong scaling) through the
    two shardings of the path that need a collective -- ``Frontend.set_depth_group`` and ``Frontend.set_view_group`` -- over
    RCCL.  Every rank runs the forward; times are the max over ranks; a second, traced pass attributes time and bytes to each
    collective (wild_deep_mvs_amd.dist.CollectiveTrace).  Returns a dict for rank 0's JSON line (None elsewhere).

    Failure handling: the set-up of a leg (model, scene: where an out-of-memory would strike) is agreed on across the ranks before
    the first collective of the leg; a failure INSIDE a collective is bounded by the process group's timeout (``rendezvous``)."""
    from wild_deep_mvs_amd.dist import CollectiveTrace
    out = {}
    for mode, cfg in SHARDED.items():
        if only is not None and mode not in only:
            continue
        net = scene = None
        err = None
        try:
            net = _sharded_net(cfg, device)
            scene = {k: v.to(device) for k, v in synthetic.make_scene(1, cfg["V"], cfg["H"], cfg["W"], seed=5 if cfg["model"] == "mvsnet" else cfg["config"]).items()}
        except Exception as e:   # pragma: no cover
            err = f"setup: {type(e).__name__}: {e}"[:300]
        if not _all_ok(dist, device, err is None):
            out[mode] = {"error": err or "setup failed on another rank"}
            del net, scene
            torch.cuda.empty_cache()
            continue
        call = lambda: net(scene["imgs"], scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"], **cfg["kw"])
        try:
            times = {}
            for label, group in (("unsharded_replicated", None), ("sharded", dist.group.WORLD)):
                getattr(net, cfg["setter"])(group)
                with torch.no_grad():
                    call(); call()
                    dist.barrier(); torch.cuda.synchronize()
                    gc.collect(); gc.disable()
                    try:
                        t0 = time.perf_counter()
                        for _ in range(reps):
                            depth = call()["depth"]
                        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
                        dt = (time.perf_counter() - t0) / reps
                    finally:
                        gc.enable()
                tm = torch.tensor([dt], device=device, dtype=torch.float64)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                times[label] = float(tm.item())
                if label == "unsharded_replicated":
                    ref_depth = depth.clone()
            with torch.no_grad(), CollectiveTrace() as tr:
                call()
            rel = float((depth - ref_depth).abs().mean() / ref_depth.abs().mean())
            coll = tr.summary()
            out[mode] = {"config": cfg["config"], "model": cfg["model"], "views": cfg["V"], "image": [cfg["H"], cfg["W"]], "kwargs": cfg["kw"],
                         "scaling": "strong", "ms_per_forward_1gpu": times["unsharded_replicated"] * 1e3,
                         "ms_per_forward_sharded": times["sharded"] * 1e3, "n_gpus": world,
                         "voxels_per_s": cfg["vox"] / times["sharded"], "speedup_vs_1gpu": times["unsharded_replicated"] / times["sharded"],
                         "depth_rel_l1_vs_unsharded": rel, "collectives": coll,
                         "collective_bytes_per_rank_per_forward": sum(c["bytes_per_rank"] * c["calls"] for c in coll),
                         "collective_ms_per_forward": sum(c["avg_us"] * c["calls"] for c in coll) * 1e-3,
                         "timed": "full forward() incl. 2-D feature nets, eager launches, max over ranks"}
        except Exception as e:   # pragma: no cover  (never lose the headline line to a side measurement)
            out[mode] = {"error": f"{type(e).__name__}: {e}"[:300]}
        del net, scene
        torch.cuda.empty_cache()
    return out if rank == 0 else None


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=3, help="reference views per step and GPU (each with its own 4 source views): the engine runs "
                                                          "the items of a batch as one launch per layer (1 = one view at a time, as in rounds 1-2)")
    ap.add_argument("--stagger", action="store_true", help="stream mode with warp-to-warp edges between the views (measured slower than lockstep: MVSNet.batch_stagger)")
    ap.add_argument("--batch-mode", choices=["views", "streams", "batched"], default="views",
                    help="how the views of a step are launched: 'views' (default, round 6) = one single-branch hipGraph per view, each replayed on its "
                         "own HIP stream, consecutive steps not joined (wild_deep_mvs_amd.graph.ViewPipeline); 'streams' (rounds 3-5) = ONE hipGraph "
                         "whose views are parallel branches (up to 4 views); 'batched' = one launch per layer for the whole batch on one stream, "
                         "replayed as a hipGraph")
    ap.add_argument("--no-training", action="store_true", help="skip the two MVSNet training-step timings (scripts/bench_train.py) reported under 'training'")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the full-forward() timings / rooflines of BASELINE configurations 1-5")
    ap.add_argument("--sharded-budget", type=float, default=300.0, help="N > 1: seconds the sharded legs may take before the headline line is "
                    "printed without them (a stuck collective cannot be cancelled)")
    ap.add_argument("--no-sharded", action="store_true", help="N > 1: skip the depth-plane / source-view sharded legs (configurations 3 and 5)")
    ap.add_argument("--dtype", choices=sorted(DTYPES), default="bf16",
                    help="16-bit HBM storage format of the headline line (arithmetic is fp32): bf16 = the format BASELINE.json configuration 2 "
                         "names; the other format is measured too and reported under \"alt\"")
    ap.add_argument("--repeats", type=int, default=5, help="the timed region (exactly --steps steps between barrier + synchronize) is run this "
                    "many times; the line's value / ms_per_step is the MEDIAN region, min / median / max are in \"repeats\"")
    ap.add_argument("--dump-events", default=None, help="write every per-launch event duration to this file")
    ap.add_argument("--eager", action="store_true", help="time eager launches instead of hipGraph replays")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE", help="pscv_set_tuning knob (measurement runs)")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not run the two rocprofv3 counter passes that measure roofline.traffic "
                                                                    "(then the newest committed profiles/rNN_traffic.json is quoted)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend of an N > 1 run (nccl = RCCL over xGMI; gloo only "
                                                      "for the CPU rendezvous test)")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="N > 1: start the ranks, join the process group, run one all-reduce, print a one-line report and exit "
                         "(tests/test_dist_cpu.py drives this with --backend gloo on CPU)")
    return ap.parse_args(argv)


def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _spawned_rank(local_rank: int, args, port: int):
    """Entry point of one self-spawned rank: the environment torch.distributed.run would have set, then the same ``run``."""
    os.environ.update(RANK=str(local_rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(args.gpus),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this driver
    run(args)


def spawn_ranks(args):
    """``python bench.py --gpus N`` without a launcher: start the N ranks here, one process per GPU, the way the reference
    starts its own (train.py:52-59,315 and depthmap_eval.py:200: ``mp.spawn(main, nprocs=world_size)`` + a localhost TCP
    rendezvous).  Under ``python -m torch.distributed.run`` WORLD_SIZE is already set and this is skipped."""
    import torch.multiprocessing as mp
    mp.spawn(_spawned_rank, args=(args, _free_port()), nprocs=args.gpus, join=True)


def training_steps():
    """MVSNet training step (5 views 512x640, D = 192, B = 1, bf16 storage) through scripts/bench_train.py, once per 2-D extractor;
    never costs the headline line (errors are reported in place)."""
    import subprocess
    out = {}
    for fe in ("torch", "pscv"):
        try:
            r = subprocess.run([sys.executable, os.path.join(REPO, "scripts", "bench_train.py"), "--feature-engine", fe, "--steps", "5", "--warmup", "3"],
                               capture_output=True, text=True, timeout=180, cwd=REPO)
            d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
            top = sorted(d["pscv_kernels_us_per_step"].items(), key=lambda kv: -kv[1])[:5]
            out[f"extractor_{fe}"] = {"ms_per_step": d["ms_per_step"], "voxels_per_s": d["value"], "engine_kernels_ms": d["pscv_kernels_total_ms"],
                                      "top_kernels_us": dict(top)}
        except Exception as e:   # pragma: no cover
            out[f"extractor_{fe}"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    out["what"] = ("MVSNet, 5 views 512x640, D=192, B=1, bf16 storage: forward in train() + loss.backward() + Adam per step; extractor_torch = the "
                   "2-D extractor on PyTorch-ROCm autograd (default), extractor_pscv = net.feature_engine_train = 'pscv' (all views in one engine pass)")
    return out


def sharded_legs_bounded(dist, device, world, rank, budget_s, legs=None):
    """``sharded_legs`` under a wall-clock budget.  The legs are the only part of an N > 1 run that exchanges data between GPUs
    (RCCL point-to-point, reduce-scatter, all-gather); no multi-GPU node was available to exercise them before the driver's run,
    and a stuck collective cannot be cancelled -- the process group's own time-out (10 min) ends the PROCESS, headline and all.
    So the legs run in a helper thread; if it has not finished after ``budget_s`` seconds the caller gets
    ``({"error": ...}, False)``, prints the already measured headline line and leaves with ``os._exit`` (no further collective,
    no destroy: the group is wedged).  Returns ``(result, finished)``."""
    import threading
    box = {}

    def work():
        try:
            if torch.cuda.is_available() and getattr(device, "type", "cpu") == "cuda":
                torch.cuda.set_device(device)
            box["res"] = (legs or sharded_legs)(dist, device, world, rank)
        except BaseException as e:      # pragma: no cover
            box["res"] = {"error": f"{type(e).__name__}: {e}"[:300]}

    th = threading.Thread(target=work, name="pscv-sharded-legs", daemon=True)
    th.start()
    th.join(budget_s)
    if th.is_alive():
        return {"error": f"the sharded legs did not finish within {budget_s:.0f} s on rank {rank}: a collective is stuck; the headline "
                         "region (no collective in its data path) was measured before them and is unaffected"}, False
    return box.get("res"), True


def rendezvous(args):
    """(dist module or None, world, rank, local_rank) from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launcher and flag disagree)")
    if world == 1:
        return None, 1, 0, local_rank
    import datetime
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    kw = {"device_id": torch.device("cuda", local_rank)} if args.backend == "nccl" else {}
    # bounded: a rank that dies inside a collective must not leave the others waiting forever
    dist.init_process_group(args.backend, rank=rank, world_size=world, timeout=datetime.timedelta(minutes=10), **kw)
    return dist, world, rank, local_rank


def main(argv=None):
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
        return
    run(args)


def run(args):
    dist, world, rank, local_rank = rendezvous(args)
    if args.rendezvous_only:
        on_gpu = args.backend == "nccl"
        if on_gpu:
            torch.cuda.set_device(local_rank)
        x = torch.full((4,), float(rank + 1), device=torch.device("cuda", local_rank) if on_gpu else "cpu")
        if dist is not None:
            dist.all_reduce(x)
        if rank == 0:
            print(json.dumps({"rendezvous": "ok", "world": world, "backend": args.backend if dist is not None else None,
                              "allreduce": float(x[0]), "launcher": "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ
                              else ("self-spawned" if world > 1 else "none")}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    for kv in args.tune:
        from wild_deep_mvs_amd import _lib
        k, v = kv.split("=")
        _lib.set_tuning(k, int(v))
    net, sd, feats, feats_cl, proj_d, dv_d, proj, dv = build_inputs(device, rank, DTYPES[args.dtype], args.batch)
    NB = args.batch
    # "streams" (default): every view of the step on its own HIP stream, forked and joined INSIDE the captured hipGraph (parallel
    # branches of one replayed graph); "batched": the views share one launch per layer on one stream.  Every kernel is bit-stable
    # under that overlap (the LDS-staged warp kernel ships as its scalar-fp32 build; tests/test_gpu_overlap.py),
    # and the timed region ends with a bit-equality check of the replayed graph against eager launches on fresh inputs
    net.batch_streams = args.batch_mode == "streams"
    net.batch_stagger = args.stagger
    streams_mode = net.batch_streams and 2 <= NB <= net.MAX_BATCH_STREAMS
    # "views" (default since round 6): one single-branch hipGraph per view on its own stream, steps not joined (graph.ViewPipeline):
    # the views drift out of phase, 0.894-0.907 against 0.918-0.927 ms for the forked graph (profiles/r06_step_schedule.txt)
    views_mode = args.batch_mode == "views" and NB >= 2 and not args.eager

    def timed_region(dtype_name, feats_cl_, steps):
        """W warm-up steps, then exactly `steps` steps between barrier + synchronize on both sides; then the same steps once
        more eagerly with a HIP event pair around every launch (per-kernel durations)."""
        net.storage_dtype = DTYPES[dtype_name]

        def step():
            return net.hot_path(feats_cl_, proj_d, dv_d)

        with torch.no_grad():
            for _ in range(max(args.warmup, 1)):
                depth, conf = step()
            torch.cuda.synchronize()
            graph = None
            # the step is captured once and replayed; with --batch-mode streams the per-view fork / join is part of the graph (parallel
            # branches: correct on changing inputs since round 4, tests/test_gpu_mvsnet.py)
            pipe = None
            if views_mode:
                from wild_deep_mvs_amd.graph import ViewPipeline
                try:
                    pipe = ViewPipeline(net, feats_cl_, proj_d, dv_d)
                    graph = pipe
                except Exception as e:   # pragma: no cover  (keeps an N-GPU run alive)
                    print(f"[bench] per-view hipGraph capture failed ({e}); timing eager launches", file=sys.stderr)
                    pipe = graph = None
                    torch.cuda.synchronize()
            elif not args.eager:
                # the 13-launch step is launch-gap bound between its small kernels: capture it once, replay it
                try:
                    graph = torch.cuda.CUDAGraph()
                    # thread_local: the RCCL watchdog thread of a multi-GPU run must not invalidate the capture
                    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                        depth, conf = step()
                    for _ in range(2):
                        graph.replay()
                    torch.cuda.synchronize()
                except Exception as e:   # pragma: no cover  (never seen on 1 GPU; keeps an N-GPU run alive)
                    print(f"[bench] hipGraph capture failed ({e}); timing eager launches", file=sys.stderr)
                    graph = None
                    torch.cuda.synchronize()
            run = pipe.step if pipe is not None else (graph.replay if graph is not None else step)
            # no cyclic-GC pause inside a timed region (a gen-2 collection with torch loaded costs ~40 ms) -- and none right in
            # front of it either: tens of ms of host work leave the GPU idle, its clocks fall, and the first timed steps ran at the
            # ramp (20 timed steps right behind the collection: 0.79 ms per 2-view step, steady state 0.69).  So: collect first,
            # then the remaining warm-up on the path that is timed, then barrier + synchronize, then the clock starts
            gc.collect()
            gc.disable()
            for _ in range(max(args.warmup, 1) * 8):
                run()
            # the timed region -- exactly `steps` steps between barrier + synchronize on both sides -- `args.repeats` times back to back
            # (box-to-box and run-to-run spread is +-4 %: one region of 20 steps = 19 ms cannot show a 3 % change); every rank times
            # every region, the MAX over ranks is taken per region, the line reports the median region
            regions = []
            for _ in range(max(1, args.repeats)):
                if dist is not None:
                    dist.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    run()
                if pipe is not None:
                    depth, conf = pipe.results()        # join of the view streams + the last step's maps: inside the timed region
                torch.cuda.synchronize()
                if dist is not None:
                    dist.barrier()
                torch.cuda.synchronize()
                regions.append(time.perf_counter() - t0)
            # per-kernel durations: the same launches (every batch item is launched on its own: B = 1 grids), but one item after
            # the other on ONE stream, so that a kernel's HIP events time it alone rather than beside another item's kernels
            item = lambda b: net.hot_path([f[b:b + 1] for f in feats_cl_], proj_d[b:b + 1], dv_d[b:b + 1])
            with ops.EventTimer() as tm:
                t1 = time.perf_counter()
                for _ in range(steps):
                    for b in range(NB):
                        item(b)
                torch.cuda.synchronize()
                elapsed_eager = time.perf_counter() - t1
            # one view at a time (rounds 1-2's step): a replayed graph of item 0 alone
            one_view = None
            if NB > 1 and not args.eager:
                try:
                    g1 = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g1, capture_error_mode="thread_local"):
                        item(0)
                    for _ in range(max(args.warmup, 1) * 8):       # the same warm-up as the headline region: clocks settled on this path
                        g1.replay()
                    torch.cuda.synchronize()
                    t2 = time.perf_counter()
                    for _ in range(steps):
                        g1.replay()
                    torch.cuda.synchronize()
                    one_view = (time.perf_counter() - t2) / steps
                except Exception:   # pragma: no cover
                    one_view = None
            gc.enable()
            # the replayed (forked) graph on FRESH inputs against eager one-item launches on one stream, bit for bit: a fast step
            # that replays wrongly must not yield a headline (round 3 saw that with the packed warp build)
            graph_ok = None
            if graph is not None:
                saved = [f.clone() for f in feats_cl_]
                for f in feats_cl_:
                    f.copy_(torch.roll(f, shifts=(3, 5), dims=(1, 2)))
                if pipe is not None:
                    pipe.step()
                    depth, conf = pipe.results()
                else:
                    graph.replay()
                torch.cuda.synchronize()
                d_graph = depth.clone()
                # (stream mode launches B = 1 grids: compared with one-item launches on one stream, i.e. without any overlap; the
                #  batched mode picks other kernel variants at larger batches -- row-split instead of reduction-split tiles, another
                #  summation order -- so its replay is compared with the same batched launches run eagerly)
                d_eager = torch.cat([item(b)[0] for b in range(NB)], 0) if (streams_mode or views_mode or NB == 1) else step()[0].clone()
                torch.cuda.synchronize()
                graph_ok = bool(torch.equal(d_graph, d_eager))
                for f, sv in zip(feats_cl_, saved):
                    f.copy_(sv)
                if pipe is not None:
                    pipe.step()
                    depth, conf = pipe.results()
                else:
                    graph.replay()
                torch.cuda.synchronize()
        assert torch.isfinite(depth).all()
        assert graph_ok is not False, "the replayed hipGraph of the step differs from eager launches on fresh inputs"
        t_max = torch.tensor(regions, device=device, dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        regions = [float(x) for x in t_max.tolist()]
        elapsed = sorted(regions)[(len(regions) - 1) // 2]           # the median region (lower median for an even count)
        return elapsed, tm, depth.clone(), graph is not None, elapsed_eager, one_view, graph_ok, regions

    elapsed, tm, depth, graphed, elapsed_eager, one_view, graph_ok, regions = timed_region(args.dtype, feats_cl, args.steps)
    kern = {k: v for k, v in tm.summary().items() if k != "proj_cams"}
    if args.dump_events and rank == 0:
        with open(args.dump_events, "w") as f:
            for name, e0, e1 in tm.records:
                f.write(f"{name}\t{e0.elapsed_time(e1) * 1e3:.1f}\n")
    # the other 16-bit storage format on the same workload (BASELINE.json names bf16; same bytes, same MFMA rate)
    alt_name = "bf16" if args.dtype == "f16" else "f16"
    feats_alt = [ops.to_channels_last(feats[i].to(device), DTYPES[alt_name]) for i in range(V)]
    alt_elapsed, alt_tm, alt_depth, _, _, alt_one_view, _, alt_regions = timed_region(alt_name, feats_alt, args.steps)
    net.storage_dtype = DTYPES[args.dtype]
    graph = graphed

    sharded, sharded_done = None, True
    if world > 1 and not args.no_sharded:
        sharded, sharded_done = sharded_legs_bounded(dist, device, world, rank, args.sharded_budget)

    if rank == 0:
        total_ms = sum(ms for _, ms in kern.values())
        name, (n, ms) = max(kern.items(), key=lambda kv: kv[1][1])
        avg_s = ms / n * 1e-3
        ab = algorithmic_bytes(name)
        traffic, traffic_source = pmc_traffic(name)
        if world == 1 and not args.no_live_traffic:
            live, note = live_traffic(args.dtype)
            hit = [v for k, v in (live or {}).items() if name.startswith(k)]
            if hit:
                traffic, traffic_source = hit[0], note
            else:
                traffic_source = f"{traffic_source} (live capture unavailable: {note})"
        roof = {"kernel": name, "bound": "hbm", "achieved": ab / avg_s / 1e9 if ab else None, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": (ab / avg_s / 1e9 / HBM_PEAK_GBS) if ab else None,
                "traffic": traffic, "traffic_source": traffic_source,
                "avg_us": avg_s * 1e6, "algorithmic_bytes": ab, "share_of_gpu_time": ms / total_ms}
        # the matrix-core side of the path (north_star: "MFMA utilisation on the 3D conv against gfx950 peak"): the layer with
        # 68 % of the regulariser's FLOPs, conv0 32 -> 8 at full resolution (2 * 27 * 32 * 8 FLOP per voxel).  Per layer it is
        # HBM-bound at these channel counts (arithmetic intensity 173 FLOP/B < ridge), so both fractions are given.
        roof_mfma = None
        c0 = [k for k in kern if k.startswith("conv3d[32->8,k")]
        if c0:
            n0, ms0 = kern[c0[0]]
            t0s = ms0 / n0 * 1e-3
            fl = 2.0 * 27 * 32 * 8 * VOX
            roof_mfma = {"kernel": c0[0], "bound": "mfma", "achieved": fl / t0s / 1e12, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": fl / t0s / 1e12 / MFMA_PEAK_TFLOPS, "avg_us": t0s * 1e6, "flops": fl,
                         "hbm_frac": algorithmic_bytes(c0[0]) / t0s / 1e9 / HBM_PEAK_GBS}
        # how much of the sweep the rig gives away: share of the (workgroup, source view) pairs whose source box lies outside the
        # source image (the view contributes 0 there and the LDS-staged kernel skips it); the DTU-like rig of `alt_geometry` has 0.03
        zero_note = ""
        try:
            zs = staging_modes(device, DTYPES[args.dtype], "probe")[0]
            zero_note = (f" ({zs['ZERO']:.2f} of its (tile, plane chunk, source view) pairs project outside the source image and are skipped; "
                         "the DTU-like rig -- alt_geometry.dtu / value_dtu_rig -- has 0.03)")
        except Exception as e:   # pragma: no cover
            zero_note = f" (staging-mode histogram unavailable: {type(e).__name__})"
        line = {
            "metric": "cost-volume voxels/sec (BxDxHxW), MVSNet hot path", "value": world * NB * VOX * args.steps / elapsed,
            "unit": "voxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "dtype_note": "16-bit HBM storage and MFMA operands, fp32 accumulation.  bf16 is the format BASELINE.json configuration 2 names "
                          "(the default of this script since round 6); the other 16-bit format (same bytes, same MFMA rate) is measured in the "
                          "same run and reported under \"alt\" with its own depth error",
            "repeats": {"n": len(regions), "steps_per_region": args.steps, "ms_per_step_min": min(regions) / args.steps * 1e3,
                        "ms_per_step_median": elapsed / args.steps * 1e3, "ms_per_step_max": max(regions) / args.steps * 1e3,
                        "ms_per_step_all": [round(r / args.steps * 1e3, 4) for r in regions],
                        "what": "the timed region (exactly `steps` steps between barrier + synchronize on both sides, max over ranks) "
                                "run `n` times back to back; value / ms_per_step are the MEDIAN region"},
            "config": {"workload": "MVSNet variance, 1 ref + 4 src views, 512x640 images (128x160x32 features), D=192, "
                                   f"features resident in HBM -> depth + confidence; a step = a batch of {NB} reference view(s) per GPU, "
                                   "each with its own 4 source views; synthetic camera rig 'probe'" + zero_note, "global_batch": world * NB, "batch_per_gpu": NB,
                       "voxels_per_step_per_gpu": NB * VOX, "parallelism": f"reference-view shard x{world}, no collective",
                       "batch_mode": ("one view" if NB == 1 else args.batch_mode if (views_mode or args.batch_mode != "views") else
                                      "batched (eager: --batch-mode views needs graph replay)"),
                       # like-for-like with rounds 1-2 (whose step was ONE reference view): the same path, one view per replay
                       "one_view_at_a_time_ms": None if one_view is None else one_view * 1e3,
                       "one_view_at_a_time_voxels_per_s": None if one_view is None else world * VOX / one_view},
            "timing": ("hipGraph replay of the step" if graph else "eager launches") +
                      (f"; a step = one replay of each of {NB} single-branch hipGraphs (one per reference view), each on its own HIP stream; "
                       "consecutive steps are not joined, so the views drift out of phase and one view's warp runs beside another's U-Net "
                       "(wild_deep_mvs_amd.graph.ViewPipeline; the join of the streams and the last step's depth maps are inside the timed "
                       "region); a step is SHORTER than the sum of its kernels' stand-alone durations below" if views_mode else
                       f"; the {NB} views of a step run on {NB} HIP streams forked inside the replayed graph: one view's vector-ALU-bound warp "
                       "beside another's MFMA / memory-bound U-Net (MVSNet._hot_path_streams), so a step is SHORTER than the sum of its "
                       "kernels' stand-alone durations below" if streams_mode else
                       f"; the {NB} views of a step share ONE launch per layer (batched grids, one stream)" if NB > 1 else "") +
                      f"; kernels_us / roofline: HIP events of an eager pass of the same {args.steps} steps, one view after the other on one "
                      f"stream (each launch timed alone; {elapsed_eager / args.steps * 1e3:.3f} ms/step that way)",
            "one_view_at_a_time": None if one_view is None else {"ms_per_view": one_view * 1e3, "value": world * VOX / one_view, "unit": "voxels/s",
                                                                 "what": "the step of rounds 1-2: one reference view per replay on one stream"},
            "graph_replay_equals_eager_on_fresh_inputs": graph_ok,
            "value_dtu_rig": None,   # filled below: the same step on the DTU-like camera rig (alt_geometry)
            "value_bf16": None,      # filled below: the same step in bf16 storage, the format BASELINE configuration 2 names
            "roofline": roof,
            "roofline_mfma": roof_mfma,
            "kernels_us": {k: round(ms / n * 1e3, 2) for k, (n, ms) in sorted(kern.items(), key=lambda kv: -kv[1][1])},
        }
        alt_kern = {k: v for k, v in alt_tm.summary().items() if k != "proj_cams"}
        line["alt"] = {"dtype": alt_name, "value": world * NB * VOX * args.steps / alt_elapsed, "unit": "voxels/s",
                       "one_view_at_a_time_ms": None if alt_one_view is None else alt_one_view * 1e3,
                       "ms_per_step": alt_elapsed / args.steps * 1e3, "ms_per_step_all": [round(r / args.steps * 1e3, 4) for r in alt_regions],
                       "kernels_us": {k: round(ms / n * 1e3, 2) for k, (n, ms) in sorted(alt_kern.items(), key=lambda kv: -kv[1][1])}}
        line["value_bf16"] = line["alt"]["value"] if alt_name == "bf16" else line["value"]
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"], rel = cpu_baseline(sd, feats, proj, dv, {args.dtype: depth, alt_name: alt_depth})
            line["depth_rel_l1_vs_oracle"] = rel[args.dtype]
            line["alt"]["depth_rel_l1_vs_oracle"] = rel[alt_name]
        else:
            line["cpu_baseline"] = None
        # all five BASELINE configurations in their single-GPU forms (parity cases of tests/test_gpu_fullsize.py, not bench lines):
        # driver-timed ms of the full forward() called like the reference's scripts call it (in-forward hipGraph replay and eager),
        # with the three heaviest kernels of each against their rooflines; after the headline region
        line["alt_geometry"] = None
        if world == 1 and not args.no_other_configs:
            try:
                line["alt_geometry"] = geometry_probe(net, device, DTYPES[args.dtype], nb=NB, views=views_mode)
                line["value_dtu_rig"] = (line["alt_geometry"].get("dtu") or {}).get("value")
            except Exception as e:   # pragma: no cover
                line["alt_geometry"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        line["sharded"] = sharded
        line["other_configs"] = None
        if world == 1 and not args.no_other_configs:
            try:
                sys.path.insert(0, os.path.join(REPO, "scripts"))
                import run_configs
                line["other_configs"] = [run_configs.time_config(c) for c in (1, 2, 3, 4, 5)]
            except Exception as e:   # pragma: no cover  (never lose the headline line to a side measurement)
                line["other_configs"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        # row f-1 of the scope table (the backward of the path): one MVSNet training step at the headline size, forward in train() +
        # loss.backward() + Adam, with the PyTorch-ROCm 2-D extractor (the default) and with the extractor on the engine too; each in
        # its own process (scripts/bench_train.py), after the headline region
        line["training"] = None
        if world == 1 and not args.no_training and not args.no_other_configs:      # (measurement children pass --no-other-configs)
            line["training"] = training_steps()
        print(json.dumps(line), flush=True)
    if not sharded_done:        # a wedged process group: nothing more can be agreed on; every rank leaves on its own
        sys.stdout.flush()
        os._exit(0)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
