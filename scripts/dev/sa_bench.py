#!/usr/bin/env python3
"""Dev: pscv_softargmin at the headline size (fp32 logits 192 x 128 x 160, depth + 4-plane confidence): time per launch.
PSCV_LIB selects another build of the library (scripts/dev/ab_build.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wild_deep_mvs_amd import _lib as L  # noqa: E402
if os.environ.get("PSCV_LIB"):
    L.LIB_PATH = os.environ["PSCV_LIB"]
from wild_deep_mvs_amd import ops  # noqa: E402

g = torch.Generator().manual_seed(0)
logits = (torch.randn(1, 192, 128, 160, generator=g) * 3).cuda()
dv = torch.linspace(2, 6, 192).view(1, -1).cuda()
fn = lambda: ops.softargmin(logits, dv, want_conf=True, conf_mode=0)
o = fn()
ref = torch.softmax(logits.double(), 1)
d = (ref * dv.double().view(1, -1, 1, 1)).sum(1).float()
print("depth max err", float((o["depth"] - d).abs().max()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rnd in range(3):
    for _ in range(5):
        fn()
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{os.environ.get('PSCV_LIB', 'default')}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
