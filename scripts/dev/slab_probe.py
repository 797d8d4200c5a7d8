"""Single-process emulation of the slab-sharded RegFuse (no torch.distributed): is `RegFuse on [a-8, b+8)` == `RegFuse on the whole
volume` on the owned units, at configuration-5 sizes, along depth and along rows?  And how far do two 16-bit shares move the result?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wild_deep_mvs_amd import ops, synthetic
from wild_deep_mvs_amd import dist as pdist
from wild_deep_mvs_amd.models.VisMVSNet.frontend import Frontend

net = Frontend()
net.load_state_dict(synthetic.sharpened_state_dict("vis", synthetic.template_of(net), seed=0))
net = net.cuda().eval()
st = net.model.stage1
g = torch.Generator(device="cuda").manual_seed(0)
for (d, h, w) in [(32, 32, 40), (64, 72, 100), (256, 144, 200), (32, 288, 400), (16, 576, 800)]:
    n_views, world = 8, 2
    interms = [(torch.randn(1, d, h, w, 8, generator=g, device="cuda") * 0.5).to(torch.float16) for _ in range(n_views)]
    uncerts = [torch.randn(1, h, w, generator=g, device="cuda") * 0.5 for _ in range(n_views)]
    with torch.no_grad():
        fused = ops.fuse_pairs(interms, uncerts)
        score = st.reg_fuse(fused)
        full = ops.softargmin(score, None, want_index=True)["index"]
        # shares
        shares = []
        wsum = None
        parts = []
        for r in range(world):
            mine = list(range(r, n_views, world))
            p, ws = ops.fuse_pairs([interms[i] for i in mine], [uncerts[i] for i in mine], normalise=False, want_wsum=True)
            parts.append(p); wsum = ws if wsum is None else wsum + ws
        shares = [ops.fuse_finish(p, wsum, torch.float16) for p in parts]
        summed = (shares[0].float() + shares[1].float()).to(torch.float16)       # what a 16-bit reduce gives
        e_share = float((summed.float() - fused.float()).abs().max() / fused.float().abs().max())
        score_s = st.reg_fuse(summed)
        idx_s = ops.softargmin(score_s, None, want_index=True)["index"]
        print(f"(d,h,w)=({d},{h},{w}): 16-bit shares vs one rounding: fused max rel {e_share:.2e}, index rel-L1 {float((idx_s - full).abs().mean() / full.abs().mean()):.2e}")
        for axis in (1, 2):
            E = fused.shape[axis]
            S = pdist.slab_size(E, world)
            if S < 8:
                continue
            worst = 0.0
            for r in range(world):
                a, b = r * S, min(E, r * S + S)
                lo, hi = max(0, a - 8), min(E, b + 8)
                ext = summed.narrow(axis, lo, hi - lo).contiguous()
                sc = st.reg_fuse(ext)
                own = sc.narrow(axis, a - lo, b - a)
                ref = score_s.narrow(axis, a, b - a)
                worst = max(worst, float((own - ref).abs().max() / ref.abs().max()))
            print(f"    axis {axis}: slab scores vs whole-volume scores, max rel {worst:.2e}")
