"""Bisect the wrong multi-branch replays: which stage of the hot path first differs from the eager run when inputs change?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench as Bn
from wild_deep_mvs_amd import ops
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
B = 2
net, sd, feats, feats_cl, proj_d, dv_d, proj, dv = Bn.build_inputs(dev, 0, torch.float16, B)
static = [f.clone() for f in feats_cl]
gen = torch.Generator(device="cuda").manual_seed(0)
ly = None


def stages(fs, b, upto):
    fb = [f[b:b + 1] for f in fs]
    cams = ops.proj_cams_device(proj_d[b:b + 1].to(torch.float32).contiguous(), 0)
    out = {"cams": cams}
    cost = net.build_cost_volume(fb[0], fb[1:], proj_d[b:b + 1, 0], [proj_d[b:b + 1, i] for i in range(1, 5)], dv_d[b:b + 1].contiguous(), cams)
    out["cost"] = cost
    if upto == "cost":
        return out
    L_ = net.cost_regularization.engine_layers(cost.dtype)
    c0 = ops.conv3d(cost, L_["conv0"]); out["c0"] = c0
    if upto == "c0":
        return out
    c1 = ops.conv3d(c0, L_["conv1"]); out["c1"] = c1
    c2 = ops.conv3d(c1, L_["conv2"]); out["c2"] = c2
    if upto == "c2":
        return out
    taps = {}
    logits, o = net.cost_regularization(cost, taps, regress=dv_d[b:b + 1].contiguous())
    out.update({k: v for k, v in taps.items()}); out["logits"] = logits; out["depth"] = o["depth"]
    return out


with torch.no_grad():
    for upto in ("cost", "c0", "c2", "all"):
        for b in range(B):
            stages(static, b, upto)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        side = [torch.cuda.Stream() for _ in range(B)]
        with torch.cuda.graph(g):
            main = torch.cuda.current_stream()
            outs = [None] * B
            for b in range(B):
                side[b].wait_stream(main)
                with torch.cuda.stream(side[b]):
                    outs[b] = stages(static, b, upto)
            for b in range(B):
                main.wait_stream(side[b])
        worst = {}
        for rep in range(6):
            for f in static:
                f.copy_((torch.randn(f.shape, generator=gen, device="cuda") * 0.5).to(f.dtype))
            refs = [stages(static, b, upto) for b in range(B)]
            torch.cuda.synchronize()
            g.replay(); torch.cuda.synchronize()
            for b in range(B):
                for k in refs[b]:
                    e = float((outs[b][k].float() - refs[b][k].float()).abs().max())
                    worst[k] = max(worst.get(k, 0.0), e)
        print(f"branches up to {upto}: max abs diff vs eager per tensor:", {k: f"{v:.2e}" for k, v in worst.items()})
