"""Dev probe: one process, emulate the depth-plane shard of Vis stage 1 and compare owned slices with the unsharded scores."""
import torch
from wild_deep_mvs_amd import synthetic, ops, dist as pdist
from wild_deep_mvs_amd.models.VisMVSNet.frontend import Frontend

net = Frontend()
net.load_state_dict(synthetic.sharpened_state_dict("vis", synthetic.template_of(net), seed=0))
net = net.cuda().eval()
kw = dict(depth_nums=[96, 16, 8], interval_scales=[1.0, 2.0, 1.0])
scene = {k: v.cuda() for k, v in synthetic.make_scene(1, 3, 64, 96, seed=1).items()}
cap = {}
st = net.model.stage1
h = st.register_forward_pre_hook(lambda m, a, k: cap.update(args=a, kwargs=k), with_kwargs=True)
net(scene["imgs"], scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"], **kw)
h.remove()
sample, k = cap["args"][0], cap["kwargs"]
D = k["depth_num"]
print("kwargs", {a: (tuple(b.shape) if torch.is_tensor(b) else b) for a, b in k.items()}, "D", D)
ref_feat, ref_cam, srcs_feat, srcs_cam = sample
ds = ref_cam[:, 1:2, 3:4, 0:1] if k.get("depth_start_override") is None else k["depth_start_override"]
di = ref_cam[:, 1:2, 3:4, 1:2] if k.get("depth_interval_override") is None else k["depth_interval_override"]
ss = k.get("s_scale", 1)
costs = st.build_cost_volume(ref_feat, ref_cam, srcs_feat, srcs_cam, D, ds, di, ss)
interm = st.reg(costs[0]); score = st.reg_pair(interm)
for rank in range(2):
    a, b = pdist.plane_shard(D, 2, rank, multiple=2)
    ea, eb = max(0, a - 16), min(D, b + 16)
    c2 = st.build_cost_volume(ref_feat, ref_cam, srcs_feat, srcs_cam, eb - ea, ds + di * ea, di, ss)
    print(rank, (a, b, ea, eb), "cost diff", (c2[0].float() - costs[0][:, ea:eb].float()).abs().max().item())
    i2 = st.reg(c2[0]); s2 = st.reg_pair(i2)
    d_i = (i2.float() - interm[:, ea:eb].float()).abs().amax(dim=(0, 2, 3, 4))
    d_s = (s2 - score[:, ea:eb]).abs().amax(dim=(0, 2, 3))
    print("  interm diff per ext plane", [f"{x:.1e}" for x in d_i.tolist()])
    print("  score  diff per ext plane", [f"{x:.1e}" for x in d_s.tolist()])

# fused part of stage 1
def stage_scores(c, lo, hi):
    ims, uns = [], []
    for i in range(len(c)):
        im = st.reg(c[i]); sc = st.reg_pair(im)
        o = ops.softargmin(score_full[i], None, want_index=True, want_entropy=True)      # global entropy
        hd = st.uncert_net(o["entropy"].unsqueeze(1))
        ims.append(im); uns.append(hd[0].squeeze(1).float().contiguous())
    fz = ops.fuse_pairs(ims, uns)
    return fz, st.reg_fuse(fz)
score_full = [st.reg_pair(st.reg(c)) for c in costs]
fz, sf = stage_scores(costs, 0, D)
for rank in range(2):
    a, b = pdist.plane_shard(D, 2, rank, multiple=2)
    ea, eb = max(0, a - 16), min(D, b + 16)
    c2 = st.build_cost_volume(ref_feat, ref_cam, srcs_feat, srcs_cam, eb - ea, ds + di * ea, di, ss)
    fz2, sf2 = stage_scores(c2, ea, eb)
    print("  fused score diff per ext plane", [f"{x:.1e}" for x in (sf2 - sf[:, ea:eb]).abs().amax(dim=(0, 2, 3)).tolist()])

# sensitivity of the fused depth to a 1e-6 relative perturbation of the pair uncertainties (unsharded)
def fused_depth(eps):
    ims, uns = [], []
    for i in range(len(costs)):
        im = st.reg(costs[i]); sc = st.reg_pair(im)
        o = ops.softargmin(sc, None, want_index=True, want_entropy=True)
        hd = st.uncert_net((o["entropy"] * (1 + eps * (1 if i == 0 else -1))).unsqueeze(1))
        ims.append(im); uns.append(hd[0].squeeze(1).float().contiguous())
    fz = ops.fuse_pairs(ims, uns)
    o = ops.softargmin(st.reg_fuse(fz), None, want_index=True)
    return o["index"], fz
i0, f0 = fused_depth(0.0)
for eps in (1e-7, 1e-6, 1e-5):
    i1, f1 = fused_depth(eps)
    print(f"eps {eps:.0e}: fused voxels changed {((f1 != f0).float().mean().item()):.2e}, index mean abs diff {(i1 - i0).abs().mean().item():.2e} "
          f"max {(i1 - i0).abs().max().item():.2e}  (index mean {i0.mean().item():.1f})")
