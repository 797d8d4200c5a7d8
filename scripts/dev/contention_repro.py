#!/usr/bin/env python3
"""Dev: is the MVSNet eval forward reproducible when ANOTHER process uses the same GPU?  P processes (no torch.distributed) loop the
forward at the headline size with taps and compare every intermediate with its own first-iteration value bit for bit; the first
tensor (in dataflow order) that differs names the launch at fault.
Usage: python scripts/dev/contention_repro.py [--procs 2] [--iters 30] [--size 512x640x192]"""
import argparse
import os
import sys

import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ORDER = ("features", "cost_volume", "conv0", "conv2", "conv4", "conv6", "up7", "up9", "up11", "logits", "depth")


def worker(rank, iters, size, q, tune, notaps=False, bar=None, views=5, partner="forward", stop=None):
    from wild_deep_mvs_amd import _lib as L
    if os.environ.get("PSCV_LIB"):
        L.LIB_PATH = os.environ["PSCV_LIB"]      # A/B runs against another build of the library
    from wild_deep_mvs_amd import synthetic
    from wild_deep_mvs_amd.models.MVSNet.model import MVSNet
    for kv in tune:
        k, v = kv.split("=")
        L.set_tuning(k, int(v))
    H, W, D = size
    dev = torch.device("cuda", 0)
    net = MVSNet("variance")
    net.load_state_dict(synthetic.sharpened_state_dict("mvsnet", synthetic.template_of(net), seed=0))
    net = net.to(dev).eval()
    net.num_depth = D
    scene = {k: v.to(dev) for k, v in synthetic.make_scene(1, views, H, W, seed=7).items()}
    first, lines, hist = {}, [], {}
    with torch.no_grad():
        net(scene["imgs"], scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"])      # weights packed, library loaded
    torch.cuda.synchronize()
    if rank > 0 and partner != "forward":
        # partner process: only one part of the forward, in a loop, until the victim (proc 0) is done
        from wild_deep_mvs_amd import ops
        from wild_deep_mvs_amd.models.MVSNet.model import build_proj_matrices
        with torch.no_grad():
            feats = net.extract_features_cl([scene["imgs"][:, i] for i in range(views)])
            sk = scene["K"].clone(); sk[:, :, :2] /= 4
            proj = build_proj_matrices(sk, scene["R"], scene["t"])
            steps = torch.arange(D, device=dev, dtype=torch.float32).view(1, 1, -1)
            dv = (scene["depth_min"].unsqueeze(-1) + ((scene["depth_max"] - scene["depth_min"]) / (D - 1)).unsqueeze(-1) * steps)[:, 0].float().contiguous()
            cams = ops.proj_cams_device(proj.float().contiguous(), 0)
            cost = net.build_cost_volume(feats[0], feats[1:], None, None, dv, cams)
            torch.cuda.synchronize()
            bar.wait()
            n = 0
            while not stop.is_set():
                for _ in range(10):
                    if partner == "warp":
                        net.build_cost_volume(feats[0], feats[1:], None, None, dv, cams)
                    elif partner == "reg":
                        net.cost_regularization(cost, None, regress=dv)
                    else:
                        net.extract_features_cl([scene["imgs"][:, i] for i in range(views)])
                torch.cuda.synchronize()
                n += 10
        q.put([f"proc {rank}: partner '{partner}' ran {n} launches-groups"])
        return
    if bar is not None:
        bar.wait()                  # the processes must OVERLAP on the GPU: start the loops together
    for it in range(iters):
        taps = {}
        with torch.no_grad():
            if notaps:
                net.graph_replay = False
                out = net(scene["imgs"], scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"])
                junk = torch.empty(int(torch.randint(1, 1 << 22, (1,))), device=dev).normal_()      # perturb the allocator's free lists
                del junk
            else:
                out = net(scene["imgs"], scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"], taps=taps)
                feats = net.extract_features_cl([scene["imgs"][:, i] for i in range(views)])
                taps["features"] = torch.stack(feats)
        taps["depth"] = out["depth"]
        torch.cuda.synchronize()
        bad = []
        for k in (("depth",) if notaps else ORDER):
            x = taps[k].detach().clone()
            if k not in first:
                first[k] = x
            elif not torch.equal(first[k].view(torch.uint8), x.view(torch.uint8)):
                d = (first[k].float() - x.float()).abs()
                nz = (d > 0).nonzero()
                ext = " ".join(f"{nz[:, a].min().item()}-{nz[:, a].max().item()}" for a in range(nz.shape[1]))
                bad.append(f"{k} ({int((d > 0).sum())} elems, max {float(d.max()):.2e}, index ranges {ext})")
                if k == "cost_volume" and len(lines) < 6:
                    vox = (d > 0).any(-1)[0]                      # [D,h,w]
                    nzv = vox.nonzero()
                    tiles = {(int(y) // 4, int(xx) // 8) for _, y, xx in nzv.tolist()}
                    planes = sorted({int(dd) for dd, _, _ in nzv.tolist()})
                    cg = [(int((d[..., 8 * g:8 * g + 8] > 0).any(-1).sum())) for g in range(4)]
                    per_tile = {}
                    for dd, y, xx in nzv.tolist():
                        per_tile.setdefault((y // 4, xx // 8), []).append((dd, y % 4, xx % 8))
                    some = list(per_tile.items())[:2]
                    det = []
                    ref_, cur_ = first[k].float()[0], x.float()[0]
                    for dd, y, xx in nzv.tolist()[:6]:
                        ch = (ref_[dd, y, xx] != cur_[dd, y, xx]).nonzero().flatten().tolist()
                        for c in ch[:3]:
                            nb = {o: round(float(ref_[dd + o, y, xx, c]), 4) for o in (-4, -2, -1, 1, 2, 4) if 0 <= dd + o < ref_.shape[0]}
                            det.append(f"(d{dd} y{y} x{xx} ch{ch}: c{c} good {float(ref_[dd, y, xx, c]):.4f} bad {float(cur_[dd, y, xx, c]):.4f} good@d+o {nb})")
                    lines.append("   detail: " + " ".join(det)[:1500])
                    lines.append(f"   structure: {int(vox.sum())} voxels in {len(tiles)} tiles, planes {planes[:24]}{'...' if len(planes) > 24 else ''}, "
                                 f"voxels with a bad channel group 0..3: {cg}; nan {int(torch.isnan(x.float()).sum())}; examples {some}"[:900])
        if bad:
            hist[bad[0].split(" ")[0]] = hist.get(bad[0].split(" ")[0], 0) + 1
            if len(lines) < 12:
                lines.append(f"proc {rank} it {it}: first differing: {bad[0]}; all: {[b.split(' ')[0] for b in bad]}")
    lines.append(f"proc {rank}: {iters} iterations, first-differing histogram {hist}")
    if stop is not None and rank == 0:
        stop.set()
    q.put(lines)


def load(stop, bar, kind):
    """Unrelated work on the same GPU from another process: fp32 matmuls (matrix pipe) or elementwise chains (vector ALU)."""
    x = torch.randn(4096, 4096, device="cuda")
    bar.wait()
    while not stop.is_set():
        for _ in range(20):
            x = (x @ x).tanh() if kind == "matmul" else (x * 1.0001 + 0.5).sin()
        torch.cuda.synchronize()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--load", choices=["matmul", "elementwise"], default=None, help="one more process with unrelated GPU work")
    ap.add_argument("--procs", type=int, default=2)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--size", default="512x640x192")
    ap.add_argument("--tune", nargs="*", default=[])
    ap.add_argument("--views", type=int, default=5)
    ap.add_argument("--partner", choices=["forward", "warp", "reg", "features"], default="forward",
                    help="what the processes other than proc 0 run: the whole forward, or one part of it in a loop")
    ap.add_argument("--no-taps", action="store_true", help="the plain forward (fused tail, intermediates freed as it goes); only the depth map is compared")
    args = ap.parse_args()
    size = tuple(int(v) for v in args.size.split("x"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    bar = ctx.Barrier(args.procs + (1 if args.load else 0))
    stop = ctx.Event()
    lp = ctx.Process(target=load, args=(stop, bar, args.load)) if args.load else None
    if lp:
        lp.start()
    procs = [ctx.Process(target=worker, args=(r, args.iters, size, q, args.tune, args.no_taps, bar, args.views, args.partner, stop)) for r in range(args.procs)]
    for p in procs:
        p.start()
    for _ in procs:
        print("\n".join(q.get(timeout=900)), flush=True)
    for p in procs:
        p.join(60)
    if lp:
        stop.set()
        lp.join(30)


if __name__ == "__main__":
    main()
