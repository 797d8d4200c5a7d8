#!/usr/bin/env python3
"""Do the two training paths of MVSNet learn alike?  N Adam steps on ONE synthetic sample (5 views 256x320, D=96) from the same
initial weights, once with the PyTorch-ROCm 2-D extractor (fp32) and once with the extractor on the engine (16-bit activations,
all views in one grouped pass); prints the loss every few steps for both storage formats."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wild_deep_mvs_amd import synthetic
from wild_deep_mvs_amd.models.MVSNet.model import MVSNet

H, W, V, D, N = 256, 320, 5, 96, 40
scene = synthetic.make_scene(1, V, H, W, seed=0)
gt, mask = synthetic.train_target(scene, H // 4, W // 4)
dev = {k: v.cuda() for k, v in scene.items() if isinstance(v, torch.Tensor)}
gt, mask = gt.cuda(), mask.cuda()
for dt in (torch.bfloat16, torch.float16):
    for fe in ("torch", "pscv"):
        net = MVSNet("variance")
        net.load_state_dict(synthetic.train_state_dict("mvsnet", synthetic.template_of(net), seed=0))
        net = net.cuda().train()
        net.num_depth, net.train_storage_dtype, net.feature_engine_train = D, dt, fe
        opt = torch.optim.Adam(net.parameters(), lr=1e-3)
        losses = []
        for it in range(N):
            opt.zero_grad(set_to_none=True)
            out = net(dev["imgs"], dev["K"], dev["R"], dev["t"], dev["depth_min"], dev["depth_max"])
            loss = synthetic.supervised_loss(out["depth"], gt, mask, dev["depth_min"], dev["depth_max"])
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        print(f"{str(dt).split('.')[-1]:8s} extractor {fe:5s}: " + "  ".join(f"{losses[i]:.3f}" for i in range(0, N, 5)) + f"  -> {losses[-1]:.3f}", flush=True)
