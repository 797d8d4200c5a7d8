"""The reference's training harness methods on the mirror: ``models.trainer.Trainer.step`` / ``test`` / ``log_iter`` /
``log_epoch`` driven the way ``train.py:185-230`` drives them, with this package installed as ``models``
(``/root/reference/models/trainer.py:61-207,280-321``)."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import wild_deep_mvs_amd
    wild_deep_mvs_amd.install_as_models()
    from wild_deep_mvs_amd import synthetic
    import models.trainer as MT
    from models.MVSNet.model import MVSNet
    return synthetic, MT, MVSNet


def _args(**kw):
    base = dict(architecture="mvsnet", upsample_training=False, occ_masking=False, supervised=False, num_im_train=3,
                print_every=1, dataset="dtu_yao", geom_clamping=0.01)
    base.update(kw)
    return types.SimpleNamespace(**base)


def _sample(synthetic, V=3, H=64, W=96, seed=2):
    scene = synthetic.make_scene(1, V, H, W, seed=seed)
    depth = (0.5 * (scene["depth_min"][:, :1] + scene["depth_max"][:, :1])).view(1, 1, 1, 1).expand(1, 1, H, W).clone()
    depth = depth * (1.0 + 0.1 * torch.rand(1, 1, H, W, generator=torch.Generator().manual_seed(0)))
    mask = torch.ones(1, 1, H, W)
    mask[..., :4, :] = 0
    return dict(scene, depth=depth, mask=mask)


@pytest.mark.parametrize("supervised", [False, True])
def test_trainer_step_backward_and_logging(env, supervised):
    synthetic, MT, MVSNet = env
    net = MVSNet("variance")
    net.load_state_dict(synthetic.sharpened_state_dict("mvsnet", synthetic.template_of(net), seed=0))
    net.num_depth = 16
    net = net.cuda().train()
    tr = MT.Trainer(net, _args(supervised=supervised))
    sample = _sample(synthetic)
    loss = tr.step(sample, True)
    assert loss.dim() == 0 and torch.isfinite(loss) and float(loss) > 0
    loss.backward()
    grads = [p.grad for p in net.parameters() if p.grad is not None]
    assert len(grads) > 50 and all(torch.isfinite(g).all() for g in grads) and any(float(g.abs().max()) > 0 for g in grads)
    assert "ref_img" in tr.ims and "scale_0_depth_est" in tr.ims and tr.ims["scale_0_depth_est"].shape[1] == 3
    if not supervised:
        assert any(k.startswith("warped") for k in tr.ims)        # photometricloss logged its warped images
    assert tr.nb_iter == 1 and abs(float(tr.log_iter()["train_loss"]) - float(loss)) < 1e-6
    val = tr.step(sample, False)
    assert set(tr.loss_means) == {"train_loss", "val_loss"} and torch.isfinite(val)


def test_trainer_test_metrics_and_log_epoch(env):
    import os
    import torch.distributed as dist
    synthetic, MT, MVSNet = env
    from models.utils import AbsDepthError_metrics
    net = MVSNet("variance")
    net.load_state_dict(synthetic.sharpened_state_dict("mvsnet", synthetic.template_of(net), seed=0))
    net.num_depth = 16
    net = net.cuda().eval()
    tr = MT.Trainer(net, _args())
    sample = _sample(synthetic)
    tr.test(sample)
    assert set(tr.loss_means) == {"EPE", "1pxError", "3pxError"} and tr.nb_iter == 1
    with torch.no_grad():
        out = net(sample["imgs"].cuda(), sample["K"].cuda(), sample["R"].cuda(), sample["t"].cuda(), sample["depth_min"].cuda(),
                  sample["depth_max"].cuda())
        step = ((sample["depth_max"] - sample["depth_min"]) / 128)[:, 0].cuda()
        est = torch.nn.functional.interpolate(out["depth"].unsqueeze(1), sample["mask"].shape[-2:], mode="bilinear",
                                              align_corners=False).squeeze(1) / step
        want = AbsDepthError_metrics(est, sample["depth"][:, 0].cuda() / step, sample["mask"][:, 0].cuda() > 0.5)
    assert abs(float(tr.loss_means["EPE"]) - float(want)) < 1e-5 * max(1.0, float(want))
    assert 0.0 <= float(tr.loss_means["3pxError"]) <= float(tr.loss_means["1pxError"]) <= 1.0
    # log_epoch averages over the ranks of the default group (train.py:236-238): one rank here
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("gloo", rank=0, world_size=1)
        created = True
    try:
        res = tr.log_epoch(7)
    finally:
        if created:
            dist.destroy_process_group()
    assert res["epoch"] == 7 and abs(float(res["EPE"]) - float(want)) < 1e-5 * max(1.0, float(want)) and tr.nb_iter == 0


@pytest.mark.parametrize("architecture", ["mvsnet", "mvsnet-s", "vis_mvsnet", "cvp_mvsnet"])
def test_load_network_flow_dataparallel_module_prefix_strict_false(env, architecture, tmp_path):
    """The reference's evaluation loader, step by step (evaluation/pipeline_utils.py:127-160): ``torch.load`` of a train.py
    checkpoint (``{'epoch','model','optimizer','architecture'}``, DDP-prefixed ``module.`` keys, train.py:205-210) -> the model
    class by architecture name through the ``models.*`` import paths -> attribute overrides -> ``nn.DataParallel`` -> ``.to(device)``
    -> ``load_state_dict(strict=False)`` -> ``eval()`` -> ``net(...)``.  Every tensor of the checkpoint must land (strict=False
    would silently drop a misnamed key) and the wrapped forward must equal the bare model's."""
    import torch.nn as nn
    synthetic, MT, MVSNet = env
    from models.VisMVSNet.frontend import Frontend as Vis_MVSNet          # the reference's names (pipeline_utils.py:22-24)
    from models.CVP_MVSNet.frontend import Frontend as CVP_MVSNet
    key = {"mvsnet": "mvsnet", "mvsnet-s": "mvsnet", "vis_mvsnet": "vis", "cvp_mvsnet": "cvp"}[architecture]
    make = {"mvsnet": lambda: MVSNet(aggregation="variance"), "mvsnet-s": lambda: MVSNet(aggregation="softmin"),
            "vis_mvsnet": Vis_MVSNet, "cvp_mvsnet": CVP_MVSNet}[architecture]
    # --- what train.py wrote
    trained = make()
    sd = synthetic.sharpened_state_dict(key, synthetic.template_of(trained), seed=3)
    if architecture == "mvsnet-s":
        sd["temp"] = torch.tensor([1.7])
    ckpt = tmp_path / "model_000005.ckpt"
    torch.save({"epoch": 5, "model": {"module." + k: v for k, v in sd.items()}, "optimizer": {}, "architecture": architecture}, ckpt)
    # --- load_network
    loaded = torch.load(ckpt)
    net = make()
    if architecture == "cvp_mvsnet":
        net.model.nscale = 2
    elif architecture == "vis_mvsnet":
        net.depth_nums, net.interval_scales = [16, 8, 4], [2, 1, 0.5]
    net = nn.DataParallel(net)
    net.to(torch.device("cuda"))
    res = net.load_state_dict(loaded["model"], strict=False)
    assert list(res.missing_keys) == [] and list(res.unexpected_keys) == [], res
    net.eval()
    for k, v in net.module.state_dict().items():
        assert torch.equal(v.cpu(), sd[k]), k
    scene = synthetic.make_scene(1, 3, 64, 96, seed=6)
    if architecture.startswith("mvsnet"):
        net.module.num_depth = 16
    args = [scene[k].cuda() for k in ("imgs", "K", "R", "t", "depth_min", "depth_max")]
    with torch.no_grad():
        out = net(*args)
        bare = net.module(*args)
    assert set(out) == {"depth", "depth_est_list", "depth_pair_list", "photometric_confidence"}
    assert torch.isfinite(out["depth"]).all() and torch.equal(out["depth"], bare["depth"])
    down = {"mvsnet": 4, "mvsnet-s": 4, "vis_mvsnet": 2, "cvp_mvsnet": 1}[architecture]           # pipeline_utils.py:140-154
    assert tuple(out["depth"].shape) == (1, 64 // down, 96 // down)
