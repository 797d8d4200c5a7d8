#!/usr/bin/env python3
"""Dev: a few eager full forwards of BASELINE configuration 2 (MVSNet 5 x 512x640, D = 192) for `rocprofv3 --kernel-trace --stats`:
which kernels -- engine and PyTorch glue -- a caller's forward() launches.  python scripts/dev/fwd_kernels.py [n]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wild_deep_mvs_amd import synthetic  # noqa: E402
from wild_deep_mvs_amd.models.MVSNet.model import MVSNet  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
net = MVSNet("variance")
net.load_state_dict(synthetic.sharpened_state_dict("mvsnet", synthetic.template_of(net), seed=0))
net = net.cuda().eval()
net.num_depth = 192
net.graph_replay = False
sc = {k: v.cuda() for k, v in synthetic.make_scene(1, 5, 512, 640, seed=2).items()}
with torch.no_grad():
    for _ in range(n):
        net(sc["imgs"], sc["K"], sc["R"], sc["t"], sc["depth_min"], sc["depth_max"])
torch.cuda.synchronize()
