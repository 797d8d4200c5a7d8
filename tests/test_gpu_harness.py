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
