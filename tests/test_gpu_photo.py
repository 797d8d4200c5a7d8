"""Unsupervised photometric loss on the HIP kernels against the reference's goldens and the CPU oracle."""
import os

import numpy as np
import pytest
from _util import retry_infra
import torch

from oracle import photometric as P
from wild_deep_mvs_amd import synthetic
from wild_deep_mvs_amd.models.trainer import Trainer
from wild_deep_mvs_amd.utils.ssimLoss import SSIM

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load_case(tag):
    g = np.load(os.path.join(GOLD, f"{tag}.npz"))
    B, V, H, W, seed, behind, masked, i_ref = [int(v) for v in g["meta"]]
    sc = synthetic.make_photo_case(B, V, H, W, seed=seed, behind_view=behind)
    return g, sc, torch.from_numpy(g["proj"]), bool(masked), i_ref


def masked_mean(ssim, mask):
    mask = mask.float()
    return torch.sum(ssim * mask) / torch.sum(mask)


@pytest.mark.parametrize("tag", ["photo_tiny", "photo_behind"])
def test_photometricloss_matches_reference_golden(tag):
    g, sc, proj, _, _ = load_case(tag)
    tr = Trainer()
    depth = sc["depths"][0].cuda().requires_grad_(True)
    ssim, mask = tr.photometricloss(sc["imgs"].cuda(), depth, proj.cuda())
    loss = masked_mean(ssim, mask)
    loss.backward()
    moved = (mask.cpu().numpy() != g["mask"]).mean()
    err = np.abs(ssim.detach().cpu().numpy() - g["ssim"])
    rel = np.abs(depth.grad.cpu().numpy() - g["grad_depth"]).sum() / np.abs(g["grad_depth"]).sum()
    print(f"[parity] {tag}: ssim max abs {err.max():.2e}, mask pixels differing {moved:.2e}, loss {float(loss):.6f} vs "
          f"{float(g['loss']):.6f}, grad_depth rel-L1 {rel:.2e}", flush=True)
    assert moved <= 1e-3                        # |g| < 1 is a threshold on an fp32 coordinate
    assert np.quantile(err, 0.999) <= 1e-4 and abs(float(loss) - float(g["loss"])) <= 1e-5
    assert rel <= 1e-4
    assert sorted(tr.ims) == [f"warped{i}" for i in range(1, sc["imgs"].shape[1])]


def test_ssim_module_matches_oracle_and_autograd():
    g = torch.Generator().manual_seed(3)
    a, b = torch.rand(2, 3, 37, 53, generator=g), torch.rand(6, 3, 37, 53, generator=g)
    bc = b.clone().requires_grad_(True)
    want = P.ssim_loss(a.repeat_interleave(3, dim=0), bc)
    go = torch.rand(want.shape, generator=g)
    want.backward(go)
    bg = b.cuda().requires_grad_(True)
    got = SSIM()(a.cuda(), bg)
    got.backward(go.cuda())
    assert (got.detach().cpu() - want.detach()).abs().max().item() <= 2e-5
    rel = ((bg.grad.cpu() - bc.grad).abs().sum() / bc.grad.abs().sum()).item()
    print(f"[parity] SSIM: value max abs {(got.detach().cpu() - want.detach()).abs().max().item():.2e}, grad rel-L1 {rel:.2e}", flush=True)
    assert rel <= 1e-4


def test_get_flow_from_depthmap_matches_oracle():
    g, sc, proj, _, _ = load_case("photo_behind")
    h, w = sc["depths"].shape[-2:]
    want_f, want_z = P.get_flow_from_depthmap(sc["depths"][0], proj, (h, w), 0)
    got_f, got_z = Trainer().get_flow_from_depthmap(sc["depths"][0].cuda(), proj.cuda(), (h, w), 0)
    assert (got_z.cpu() - want_z).abs().max().item() <= 1e-4
    assert ((got_f.cpu() - want_f).abs() > 1e-4).float().mean().item() <= 1e-4       # z ~ 0 pixels may flip to -10
    assert (want_f == -10).float().mean().item() > 0.2


def _masked_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from argparse import Namespace
        g, sc, proj, _, i_ref = load_case("photo_masked")
        tr = Trainer(args=Namespace(occ_masking=True, geom_clamping=float(g["geom_clamping"])))
        depth = sc["depths"][rank].cuda().requires_grad_(True)            # rank r predicts view r
        ssim, mask = tr.loss(sc["imgs"].cuda(), depth, proj.cuda(), None)
        loss = masked_mean(ssim, mask)
        loss.backward()
        q.put((rank, ssim.detach().cpu().numpy(), mask.cpu().numpy(), float(loss), depth.grad.cpu().numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@retry_infra
def test_masked_photometricloss_ranks_one_gpu():
    """Occlusion masking needs one rank per view (rank r's reference view is r): four ranks on one GPU, gloo rendezvous; rank 1
    is the one the reference golden was generated for."""
    import torch.multiprocessing as mp
    import socket
    def free_port():
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            return s_.getsockname()[1]
    g, sc, proj, _, i_ref = load_case("photo_masked")
    world = sc["imgs"].shape[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_masked_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r[0]: r for r in [q.get(timeout=240) for _ in range(world)]}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0, f"a rank process exited with code {p.exitcode}"
    _, ssim, mask, loss, grad = res[i_ref]
    moved = (mask.astype(np.float32) != g["mask"]).mean()
    rel = np.abs(grad - g["grad_depth"]).sum() / np.abs(g["grad_depth"]).sum()
    print(f"[parity] photo_masked rank {i_ref}: mask pixels differing {moved:.2e}, loss {loss:.6f} vs {float(g['loss']):.6f}, "
          f"grad_depth rel-L1 {rel:.2e}", flush=True)
    assert moved <= 2e-3 and np.quantile(np.abs(ssim - g["ssim"]), 0.999) <= 1e-4
    assert abs(loss - float(g["loss"])) <= 1e-5 and rel <= 1e-4


def test_unsupervised_mvsnet_train_step():
    """The `--unsupervised` step of the reference's trainer (models/trainer.py:96-174) on the engine end to end: MVSNet in
    train() mode -> depth at 1/4 resolution -> images resized to it, intrinsics divided by 4 -> photometric loss -> backward
    into every weight.  Against the oracle (model restatement with the engine's fp16 storage emulated + loss restatement, both
    pinned to the reference): loss value, and the cosine of the full weight-gradient vector (a random-weight BatchNorm net
    amplifies one-ulp differences layer by layer, so the whole vector is compared, as in tests/test_gpu_train.py)."""
    import torch.nn.functional as F
    from oracle import mvsnet as O
    from wild_deep_mvs_amd.models.MVSNet.model import MVSNet, build_proj_matrices
    H, W, V, D, B, seed = 64, 96, 3, 16, 2, 0
    scene = synthetic.make_scene(B, V, H, W, seed=4)

    def loss_of(depth, imgs, K, R, t, photometricloss):
        h, w = depth.shape[-2:]
        img = F.interpolate(imgs.view(-1, 3, H, W), size=(h, w), mode="bilinear", align_corners=False).view(B, V, 3, h, w)
        Ks = K.clone()
        Ks[:, :, :2] /= 4
        ssim, mask = photometricloss(img, depth, build_proj_matrices(Ks, R, t))[:2]
        return torch.sum(ssim * mask) / torch.sum(mask)

    net = MVSNet("variance")
    net.load_state_dict(synthetic.train_state_dict("mvsnet", synthetic.template_of(net), seed=seed))
    net = net.cuda().train()
    net.num_depth, net.train_storage_dtype = D, torch.float16
    dev = {k: scene[k].cuda() for k in ("imgs", "K", "R", "t", "depth_min", "depth_max")}
    out = net(*[dev[k] for k in ("imgs", "K", "R", "t", "depth_min", "depth_max")])
    loss = loss_of(out["depth"], dev["imgs"], dev["K"], dev["R"], dev["t"], Trainer().photometricloss)
    loss.backward()

    sd = synthetic.train_state_dict("mvsnet", synthetic.template_of(MVSNet("variance")), seed=seed)
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v) for k, v in sd.items()}
    o_out = O.forward(scene["imgs"], scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"], sd, num_depth=D,
                      aggregation="variance", training=True, new_stats={}, store=torch.float16)
    o_loss = loss_of(o_out["depth"], scene["imgs"], scene["K"], scene["R"], scene["t"], P.photometricloss)
    o_loss.backward()
    dot = n1 = n2 = 0.0
    for k, p in net.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        a, b = p.grad.float().cpu(), sd[k].grad.float()
        dot += float((a * b).sum()); n1 += float((a * a).sum()); n2 += float((b * b).sum())
    cos = dot / (n1 ** 0.5 * n2 ** 0.5)
    print(f"[parity] unsupervised MVSNet step: loss {float(loss):.6f} vs oracle {float(o_loss):.6f}, gradient cosine {cos:.5f}", flush=True)
    assert abs(float(loss) - float(o_loss)) <= 1e-3 * abs(float(o_loss))
    assert cos >= 0.98


def test_unsupervised_vis_train_step():
    """The reference trainer's `--unsupervised` loss for Vis-MVSNet (models/trainer.py:154-200) on the engine end to end: three
    cascade depth maps and every pair depth upsampled to 1/2 resolution, photometric loss per map (factors 2, 1, 0.5), Bayesian pair
    loss `ssim exp(-u) + u` with the pair's uncertainty, backward into every weight.  Against the oracle pair (model restatement in
    train mode + loss restatement): loss value and the cosine of the full weight gradient."""
    import json
    from collections import OrderedDict
    import torch.nn.functional as F
    from oracle import vismvsnet as OV
    from wild_deep_mvs_amd.models.MVSNet.model import build_proj_matrices
    from wild_deep_mvs_amd.models.VisMVSNet.frontend import Frontend
    from wild_deep_mvs_amd.models.utils import rec_upsample, bayesian_version_loss
    H, W, V, B, seed = 64, 96, 3, 2, 0
    depth_nums, scales = [16, 8, 4], [8.0, 4.0, 2.0]
    scene = synthetic.make_scene(B, V, H, W, seed=4)

    def loss_of(out, imgs, K, R, t, photometricloss):
        h, w = H // 2, W // 2
        img = F.interpolate(imgs.view(-1, 3, H, W), size=(h, w), mode="bilinear", align_corners=False).view(B, V, 3, h, w)
        Ks = K.clone()
        Ks[:, :, :2] /= 2
        proj = build_proj_matrices(Ks, R, t)
        loss = 0
        for i, d in enumerate(rec_upsample(list(out["depth_est_list"]), (h, w))):
            ssim, mask = photometricloss(img, d, proj)[:2]
            loss = loss + [2, 1, 0.5][i] * torch.sum(ssim * mask) / torch.sum(mask)
        for i, pairs in enumerate(rec_upsample([[(d, u[0]) for d, u in st] for st in out["depth_pair_list"]], (h, w))):
            for j, (d, unc) in enumerate(pairs):
                idx = [0, j + 1]
                ssim, mask = photometricloss(img[:, idx], d.squeeze(1), proj[:, idx])[:2]
                loss = loss + [2, 1, 0.5][i] / (V - 1) * bayesian_version_loss(ssim, unc, mask)
        return loss

    net = Frontend()
    net.load_state_dict(synthetic.sharpened_state_dict("vis", synthetic.template_of(net), seed=seed))
    net = net.cuda().train()
    net.depth_nums, net.interval_scales = depth_nums, scales
    net.train_storage_dtype = torch.float16
    dev = {k: scene[k].cuda() for k in ("imgs", "K", "R", "t", "depth_min", "depth_max")}
    out = net(*[dev[k] for k in ("imgs", "K", "R", "t", "depth_min", "depth_max")])
    loss = loss_of(out, dev["imgs"], dev["K"], dev["R"], dev["t"], Trainer().photometricloss)
    loss.backward()

    keys = json.load(open(os.path.join(GOLD, "state_dict_keys.json")))["vis"]
    sd = synthetic.sharpened_state_dict("vis", OrderedDict((k, tuple(s)) for k, s in keys), seed=seed)
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running_" not in k else v) for k, v in sd.items()}
    with OV.train_mode({}):
        o_out = OV.forward(scene["imgs"], scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"], sd,
                           depth_nums=tuple(depth_nums), interval_scales=tuple(scales), attr_interval_scales=tuple(scales))
    o_loss = loss_of(o_out, scene["imgs"], scene["K"], scene["R"], scene["t"], P.photometricloss)
    o_loss.backward()
    dot = n1 = n2 = 0.0
    for k, p in net.named_parameters():
        if sd[k].grad is None:
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        a, b = p.grad.float().cpu(), sd[k].grad.float()
        dot += float((a * b).sum()); n1 += float((a * a).sum()); n2 += float((b * b).sum())
    cos = dot / (n1 ** 0.5 * n2 ** 0.5)
    print(f"[parity] unsupervised Vis step: loss {float(loss):.6f} vs oracle {float(o_loss):.6f}, gradient cosine {cos:.5f}", flush=True)
    assert abs(float(loss) - float(o_loss)) <= 2e-3 * abs(float(o_loss))
    assert cos >= 0.97
