import sys, torch
sys.path.insert(0, ".")
from wild_deep_mvs_amd import synthetic, _lib as L
from wild_deep_mvs_amd.models.MVSNet.model import MVSNet
dev = torch.device("cuda")
net = MVSNet("variance"); net.load_state_dict(synthetic.sharpened_state_dict("mvsnet", synthetic.template_of(net), seed=0)); net = net.to(dev).eval(); net.num_depth = 192; net.graph_replay = False
sc = {k: v.to(dev) for k, v in synthetic.make_scene(3, 5, 512, 640, seed=7).items()}
call = lambda: net(sc["imgs"], sc["K"], sc["R"], sc["t"], sc["depth_min"], sc["depth_max"])
for tiled in (1, 0):
    L.set_tuning("warp_tiled", tiled)
    with torch.no_grad():
        net.batch_streams = False
        ref = call()["depth"].clone()
        net.batch_streams = True
        bad = 0; worst = 0.0; npx = 0
        for _ in range(40):
            d = call()["depth"]
            if not torch.equal(d, ref):
                bad += 1; e = (d - ref).abs(); worst = max(worst, float(e.max() / ref.abs().max())); npx = max(npx, int((e > 0).sum()))
    print(f"warp_tiled={tiled}: stream mode differs from the sequential forward in {bad} of 40 steps (worst rel {worst:.2e}, up to {npx} of {ref.numel()} pixels)", flush=True)
