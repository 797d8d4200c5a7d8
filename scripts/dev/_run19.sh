set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s; mkdir -p $O
python - > $O/ppd_rigs.txt 2>&1 <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
import bench as Bn
from wild_deep_mvs_amd import _lib as L
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
net, *_ = Bn.build_inputs(dev, 0, torch.bfloat16, 3)
for rnd in range(2):
    for rig in ("probe", "dtu"):
        for ppd in (32, 48, 64):
            L.set_tuning("warp_ppd", ppd)
            ms = Bn.rig_step(net, dev, torch.bfloat16, rig, 3, steps=150, views=True)
            print(f"round {rnd} rig {rig} warp_ppd {ppd}: {ms:.4f} ms per step", flush=True)
L.set_tuning("warp_ppd", 0)
PY
grep -v amdgpu $O/ppd_rigs.txt
