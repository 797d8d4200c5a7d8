#!/usr/bin/env python3
"""Dev: ONE process, two HIP streams -- stream A repeats the MVSNet warp + cost launch and compares the volume with its first result
bit for bit, stream B keeps the GPU busy with the regulariser's MFMA conv launches (or the 2-D extractor's).  Companion of
contention_repro.py (where the partner is another PROCESS).  Usage: python scripts/dev/warp_vs_mfma_streams.py [--size 128x160x48]
[--views 3] [--iters 2000] [--partner reg|features|none] [--tune warp_tiled=0]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wild_deep_mvs_amd import _lib as L  # noqa: E402
if os.environ.get("PSCV_LIB"):
    L.LIB_PATH = os.environ["PSCV_LIB"]      # A/B runs against another build of the library
from wild_deep_mvs_amd import ops, synthetic  # noqa: E402
from wild_deep_mvs_amd.models.MVSNet.model import MVSNet, build_proj_matrices  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="128x160x48")
    ap.add_argument("--views", type=int, default=3)
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--partner", default="reg")
    ap.add_argument("--tune", nargs="*", default=[])
    args = ap.parse_args()
    for kv in args.tune:
        k, v = kv.split("=")
        L.set_tuning(k, int(v))
    H, W, D = (int(v) for v in args.size.split("x"))
    dev = torch.device("cuda", 0)
    net = MVSNet("variance")
    net.load_state_dict(synthetic.sharpened_state_dict("mvsnet", synthetic.template_of(net), seed=0))
    net = net.to(dev).eval()
    net.num_depth = D
    scene = {k: v.to(dev) for k, v in synthetic.make_scene(1, args.views, H, W, seed=7).items()}
    with torch.no_grad():
        feats = net.extract_features_cl([scene["imgs"][:, i] for i in range(args.views)])
        sk = scene["K"].clone(); sk[:, :, :2] /= 4
        proj = build_proj_matrices(sk, scene["R"], scene["t"])
        steps = torch.arange(D, device=dev, dtype=torch.float32).view(1, 1, -1)
        dv = (scene["depth_min"].unsqueeze(-1) + ((scene["depth_max"] - scene["depth_min"]) / (D - 1)).unsqueeze(-1) * steps)[:, 0].float().contiguous()
        cams = ops.proj_cams_device(proj.float().contiguous(), 0)
        first = net.build_cost_volume(feats[0], feats[1:], None, None, dv, cams).clone()
        cost_b = first.clone()
        torch.cuda.synchronize()
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
        layer_in = {}
        mm_a = torch.randn(4096, 4096, device=dev, dtype=torch.float16); mm_b = torch.randn(4096, 4096, device=dev, dtype=torch.float16)
        sm_logits = torch.randn(1, D, H // 4, W // 4, device=dev)
        ew = torch.randn(1 << 24, device=dev)
        if args.partner.startswith("layer:"):          # one launch of the regulariser: its input comes from a tapped forward
            taps = {}
            net.cost_regularization(cost_b, taps)
            ly = net.cost_regularization.engine_layers(cost_b.dtype)
            c1 = ops.conv3d(taps["conv0"], ly["conv1"]); c3 = ops.conv3d(taps["conv2"], ly["conv3"]); c5 = ops.conv3d(taps["conv4"], ly["conv5"])
            layer_in = {"conv0": (cost_b, None), "conv1": (taps["conv0"], None), "conv2": (c1, None), "conv3": (taps["conv2"], None),
                        "conv4": (c3, None), "conv5": (taps["conv4"], None), "conv6": (c5, None), "conv7": (taps["conv6"], taps["conv4"]),
                        "conv9": (taps["up7"], taps["conv2"]), "conv11": (taps["up9"], taps["conv0"]), "prob": (taps["up11"], None)}
            lname = args.partner.split(":")[1]
            lx, lskip = layer_in[lname]
            llayer = ly[lname]
            torch.cuda.synchronize()
        outs = [torch.empty_like(first) for _ in range(8)]
        bad = 0
        worst = 0.0
        keep = [f.clone() for f in feats] + [cams.clone(), dv.clone()]
        shown = 0
        for it in range(0, args.iters, 8):
            with torch.cuda.stream(sb):
                for _ in range(6):
                    if args.partner == "reg":
                        net.cost_regularization(cost_b, None, regress=dv)
                    elif args.partner == "features":
                        net.extract_features_cl([scene["imgs"][:, i] for i in range(args.views)])
                    elif args.partner == "matmul":
                        mm_c = mm_a @ mm_b
                    elif args.partner == "matmul32":
                        mm_c = mm_a.float() @ mm_b.float()
                    elif args.partner == "softargmin":
                        for _ in range(8):
                            ops.softargmin(sm_logits, dv, want_conf=True, conf_mode=0)
                    elif args.partner == "elementwise":
                        for _ in range(8):
                            ew = (ew * 1.0001 + 0.5).sin()
                    elif args.partner.startswith("layer:"):
                        for _ in range(4):
                            if lname == "prob":
                                ops.conv3d(lx, llayer, out_dtype=torch.float32)
                            else:
                                ops.conv3d(lx, llayer, skip=lskip)
            with torch.cuda.stream(sa):
                for o in outs:
                    ops.warp_cost(feats[0], feats[1:], cams, dv, geom=L.GEOM_PROJ, cost=L.COST_VARIANCE, out=o)
            torch.cuda.synchronize()
            for o in outs:
                if not torch.equal(o, first):
                    bad += 1
                    worst = max(worst, float((o.float() - first.float()).abs().max()))
                    if shown < 3:
                        shown += 1
                        d = (o.float() - first.float()).abs()[0]
                        vox = (d > 0).any(-1).nonzero().tolist()
                        pos = sorted({(y % 4, x % 8) for _, y, x in vox})
                        chans = sorted({int(c) % 8 for c in (d > 0).nonzero()[:, 3].tolist()})
                        same_in = all(torch.equal(a, b) for a, b in zip(keep, list(feats) + [cams, dv]))
                        per = {}
                        for dd, y, x in vox:
                            per.setdefault((y // 4, x // 8, y % 4), set()).add(dd)
                        ex = [(k, sorted(v)) for k, v in list(per.items())[:6]]
                        nch = {}
                        dn = (d > 0)
                        for dd, y, x in vox[:400]:
                            n = int(dn[dd, y, x].sum()); nch[n] = nch.get(n, 0) + 1
                        print(f"   planes per (tile row, tile col, row in tile): {ex}; bad channels per bad voxel (histogram over 400): {nch}", flush=True)
                        print(f"   bad launch: {len(vox)} voxels, in-tile (row, col) positions {pos}, channel index mod 8 {chans}, inputs unchanged {same_in}", flush=True)
    print(f"one process, partner stream '{args.partner}', tune {args.tune}: {bad} of {args.iters} warp launches differ from the first (max abs {worst:.3e})", flush=True)


if __name__ == "__main__":
    main()
