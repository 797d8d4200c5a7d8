"""Training path on the GPU, through the C ABI: every backward kernel against ATen autograd of the same op
on the same (16-bit rounded) operands, then one full train() step of the MVSNet mirror against the CPU oracle's
autograd (which tests/test_oracle_train.py pins to a training step of the reference itself).

Tolerances: the engine stores activations and gradients in 16 bits (bf16 by default in training) and accumulates in
fp32.  Kernel-level checks feed both sides identical rounded operands, so they are tight (fp32 accumulation order only);
end-to-end gradients carry ~a dozen 16-bit roundings each way and are compared in relative L2 per tensor."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _util import check_close, load_golden, t

pytestmark = pytest.mark.gpu

DT = [torch.bfloat16, torch.float16]


def _cl(x, dtype):   # NCDHW fp32 -> channels-last 16-bit on the GPU
    from wild_deep_mvs_amd import ops
    return ops.to_channels_last(x.cuda(), dtype)


def _cf(x):          # channels-last -> NCDHW fp32 on the CPU
    from wild_deep_mvs_amd import ops
    return ops.to_channels_first(x.detach()).float().cpu()


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("C", [8, 16, 32, 64])
def test_bn_stats_and_act(C, dtype):
    from wild_deep_mvs_amd import ops
    g = torch.Generator().manual_seed(C)
    x = (torch.randn(2, C, 6, 10, 12, generator=g) * 2 + 0.5).to(dtype).float()
    skip = torch.randn(2, C, 6, 10, 12, generator=g).to(dtype).float()
    sums = ops.bn_stats(_cl(x, dtype)).cpu()
    check_close("sum", sums[0], x.sum((0, 2, 3, 4)), rel_l2=1e-5)
    check_close("sumsq", sums[1], (x * x).sum((0, 2, 3, 4)), rel_l2=1e-5)
    sc, bi = torch.randn(C, generator=g), torch.randn(C, generator=g)
    for relu in (True, False):
        z = x * sc.view(1, C, 1, 1, 1) + bi.view(1, C, 1, 1, 1)
        ref = (F.relu(z) if relu else z) + skip
        got = ops.bn_act(_cl(x, dtype), sc.cuda(), bi.cuda(), relu=relu, skip=_cl(skip, dtype))
        check_close(f"bn_act relu={relu}", _cf(got), ref.to(dtype).float(), rel_l2=3e-3 if dtype == torch.bfloat16 else 4e-4)


@pytest.mark.parametrize("C,affine", [(8, True), (16, True), (64, True), (32, False)])
def test_bn_bookkeeping_kernels_match_batchnorm_module(C, affine):
    """pscv_bn_finalize / pscv_bn_bwd_coeffs (the [C]-vector bookkeeping around the reductions, one launch each) against
    nn.BatchNorm3d in train(): affine, saved mean / invstd, running statistics with the unbiased variance, num_batches_tracked; and
    the input-gradient coefficients + d gamma / d beta against autograd of the same module."""
    import torch.nn as nn
    from wild_deep_mvs_amd import ops
    g = torch.Generator().manual_seed(11 + C)
    y = (torch.randn(2, C, 4, 6, 16, generator=g) * 1.5 + 0.3).requires_grad_(True)
    ref_bn = nn.BatchNorm3d(C, affine=affine).train()
    if affine:
        with torch.no_grad():
            ref_bn.weight.copy_(torch.rand(C, generator=g) + 0.5); ref_bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
            ref_bn.running_mean.copy_(torch.randn(C, generator=g)); ref_bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    import copy
    bn = copy.deepcopy(ref_bn).cuda()
    z = ref_bn(y)
    dz = torch.randn(z.shape, generator=g)
    z.backward(dz)
    n = y.numel() // C
    yd = y.detach()
    sums = torch.stack([yd.sum((0, 2, 3, 4)), (yd * yd).sum((0, 2, 3, 4))]).cuda()
    out = ops.bn_finalize(sums, n, bn).cpu()
    mean, var = yd.mean((0, 2, 3, 4)), yd.var((0, 2, 3, 4), unbiased=False)
    gamma = ref_bn.weight.detach() if affine else torch.ones(C)
    beta = ref_bn.bias.detach() if affine else torch.zeros(C)
    check_close("scale", out[0], gamma * torch.rsqrt(var + 1e-5), rel_l2=1e-5)
    check_close("bias", out[1], beta - mean * gamma * torch.rsqrt(var + 1e-5), rel_l2=1e-5)
    check_close("mean", out[2], mean, rel_l2=1e-5)
    check_close("invstd", out[3], torch.rsqrt(var + 1e-5), rel_l2=1e-5)
    check_close("running_mean", bn.running_mean.cpu(), ref_bn.running_mean, rel_l2=1e-5)
    check_close("running_var", bn.running_var.cpu(), ref_bn.running_var, rel_l2=1e-5)
    assert int(bn.num_batches_tracked) == int(ref_bn.num_batches_tracked) == 1
    bsum = torch.stack([dz.sum((0, 2, 3, 4)), (dz * yd).sum((0, 2, 3, 4))]).cuda()
    co = ops.bn_bwd_coeffs(bsum, out[2].cuda(), out[3].cuda(), bn.weight, n).cpu()
    dy = co[0].view(1, C, 1, 1, 1) * dz + co[1].view(1, C, 1, 1, 1) * yd + co[2].view(1, C, 1, 1, 1)
    check_close("dy from the coefficients", dy, y.grad, rel_l2=2e-4)
    if affine:
        check_close("d gamma", co[3], ref_bn.weight.grad, rel_l2=1e-4)
        check_close("d beta", co[4], ref_bn.bias.grad, rel_l2=1e-5)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("C,relu", [(8, True), (16, True), (32, False), (64, True)])
def test_bn_backward(C, relu, dtype):
    """pscv_bn_bwd_reduce + pscv_bn_bwd_apply (+ the host's [C]-vector coefficient math) == autograd of
    batch_norm(training=True) -> relu on the same stored y."""
    from wild_deep_mvs_amd import ops
    g = torch.Generator().manual_seed(7 + C)
    y = (torch.randn(2, C, 4, 6, 16, generator=g) * 1.5 + 0.3).to(dtype).float().requires_grad_(True)
    dact = torch.randn(2, C, 4, 6, 16, generator=g).to(dtype).float()
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, generator=g) * 0.3).requires_grad_(True)
    z = F.batch_norm(y, None, None, gamma, beta, training=True, eps=1e-5)
    out = F.relu(z) if relu else z
    out.backward(dact)
    n = y.numel() // C
    mean = y.detach().mean((0, 2, 3, 4))
    var = y.detach().var((0, 2, 3, 4), unbiased=False)
    invstd = torch.rsqrt(var + 1e-5)
    scale = (gamma.detach() * invstd).cuda()
    bias = (beta.detach() - mean * gamma.detach() * invstd).cuda()
    ycl, dcl = _cl(y.detach(), dtype), _cl(dact, dtype)
    s = ops.bn_bwd_reduce(dcl, ycl, scale, bias, relu=relu).cpu()
    s1, s2 = s[0], invstd * (s[1] - mean * s[0])
    check_close("dbeta", s1, beta.grad, rel_l2=1e-4)
    check_close("dgamma", s2, gamma.grad, rel_l2=1e-4)
    k = gamma.detach() * invstd
    ca, cb, cc = k, -k * invstd * s2 / n, -k * s1 / n + k * invstd * mean * s2 / n
    dy = ops.bn_bwd_apply(dcl, ycl, scale, bias, ca.cuda(), cb.cuda(), cc.cuda(), relu=relu)
    check_close("dy", _cf(dy), y.grad, rel_l2=6e-3 if dtype == torch.bfloat16 else 1e-3)


@pytest.mark.parametrize("per_pixel", [False, True])
def test_softargmin_backward(per_pixel):
    from wild_deep_mvs_amd import ops
    g = torch.Generator().manual_seed(3)
    B, D, h, w = 2, 24, 10, 14
    logits = (torch.randn(B, D, h, w, generator=g) * 2).requires_grad_(True)
    depth = (torch.rand(B, D, h, w, generator=g) + torch.arange(D).view(1, D, 1, 1)) if per_pixel else \
        (torch.arange(D).float().view(1, D) * 0.3 + torch.tensor([[2.0], [3.0]]))
    dv = depth if per_pixel else depth.view(B, D, 1, 1)
    d = (F.softmax(logits, 1) * dv).sum(1)
    gd = torch.randn(B, h, w, generator=g)
    d.backward(gd)
    got = ops.softargmin_bwd(logits.detach().cuda(), depth.contiguous().cuda(), gd.cuda(), torch.float16).float().cpu()
    assert float(got[..., 1:].abs().max()) == 0.0
    check_close("dlogits", got[..., 0], logits.grad, rel_l2=1e-3)


WG = [  # (c_out_or_a, c_in_or_b, stride, transposed)   every block shape of the MVSNet / CVP / Vis regularisers
    (8, 32, 1, False), (16, 8, 2, False), (16, 16, 1, False), (32, 16, 2, False), (32, 32, 1, False), (64, 32, 2, False),
    (64, 64, 1, False), (64, 32, 2, True), (32, 16, 2, True), (16, 8, 2, True), (8, 8, 1, False), (64, 32, 1, True),
    (8, 16, 1, False),
]


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("ca,cb,stride,transposed", WG)
def test_conv3d_wgrad_and_dgrad(ca, cb, stride, transposed, dtype):
    """pscv_conv3d_wgrad == autograd's weight gradient; the adjoint layer of training._dgrad_layer on the forward conv
    kernels == autograd's input gradient.  Odd tile remainders on purpose (W = 20: partial 16-wide tiles)."""
    from wild_deep_mvs_amd import ops, training as T
    g = torch.Generator().manual_seed(ca * 100 + cb + stride)
    B, Do, Ho, Wo = 2, 4, 6, 20          # grid of the conv OUTPUT (Conv3d) / INPUT (ConvTranspose3d) = the coarse grid
    fine = (B, 0, stride * Do, stride * Ho, stride * Wo)
    if not transposed:                   # Conv3d weight [Co=ca, Ci=cb]
        w = (torch.randn(ca, cb, 3, 3, 3, generator=g) * 0.1).to(dtype).float().requires_grad_(True)
        x = torch.randn(B, cb, *fine[2:], generator=g).to(dtype).float().requires_grad_(True)
        y = F.conv3d(x, w, None, stride=stride, padding=1)
        dy = torch.randn(y.shape, generator=g).to(dtype).float()
        y.backward(dy)
        dw = ops.conv3d_wgrad(_cl(dy, dtype), _cl(x.detach(), dtype), ca=ca, cb=cb, stride=stride)
        blk = T.Block("b", "x", w.detach().cuda(), stride=stride, transposed=False)
    else:                                # ConvTranspose3d weight [Ci=ca, Co=cb]
        w = (torch.randn(ca, cb, 3, 3, 3, generator=g) * 0.1).to(dtype).float().requires_grad_(True)
        x = torch.randn(B, ca, Do, Ho, Wo, generator=g).to(dtype).float().requires_grad_(True)
        y = F.conv_transpose3d(x, w, None, stride=stride, padding=1, output_padding=stride - 1)
        dy = torch.randn(y.shape, generator=g).to(dtype).float()
        y.backward(dy)
        dw = ops.conv3d_wgrad(_cl(x.detach(), dtype), _cl(dy, dtype), ca=ca, cb=cb, stride=stride)
        blk = T.Block("b", "x", w.detach().cuda(), stride=stride, transposed=True)
    check_close("dW", dw.cpu(), w.grad, rel_l2=2e-5)
    lay = T._dgrad_layer(blk, dtype, "cuda")
    dx = ops.conv3d(_cl(dy, dtype), lay, out_dtype=torch.float32)
    check_close("dX", _cf(dx), x.grad, rel_l2=2e-5)


def _sweep_case(C, V, h, w, D, seed, behind=False):
    from wild_deep_mvs_amd import synthetic
    scene = synthetic.make_scene(1, V, 4 * h, 4 * w, seed=seed, behind_view=1 if behind else -1)
    from oracle import mvsnet as O
    proj, dv = O.mvsnet_cameras(scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"], D)
    g = torch.Generator().manual_seed(seed + 1)
    feats = [torch.randn(1, C, h, w, generator=g) * 0.5 for _ in range(V)]
    return proj, dv[:, 0].contiguous(), feats


@pytest.mark.parametrize("cost,C", [("variance", 32), ("variance_cvp", 16), ("softmin", 32), ("warp_only", 32)])
def test_warp_cost_backward_fp32(cost, C):
    """pscv_warp_cost_bwd with fp32 features / gradients == autograd through the oracle's grid_sample + cost statistic
    (zero-padded border taps, a camera with points behind it, softmin's temperature gradient)."""
    from wild_deep_mvs_amd import ops, _lib as L
    from oracle import mvsnet as O
    V, h, w, D = 4, 16, 24, 8
    proj, dv, feats = _sweep_case(C, V, h, w, D, seed=2, behind=True)
    feats = [f.requires_grad_(True) for f in feats]
    temp = torch.tensor([0.7], requires_grad=True)
    warped = [O.homo_warping(feats[i], proj[:, i], proj[:, 0], dv, (h, w)) for i in range(1, V)]
    if cost == "variance":
        vol, mode = O.variance_cost(feats[0], warped), L.COST_VARIANCE
    elif cost == "variance_cvp":
        N = V
        s = feats[0].unsqueeze(2) + sum(warped)
        sq = feats[0].unsqueeze(2) ** 2 + sum(w_ ** 2 for w_ in warped)
        vol, mode = sq / N - (s / N) ** 2, L.COST_VARIANCE_CVP
    elif cost == "softmin":
        vol, mode = O.softmin_cost(feats[0], warped, temp), L.COST_SOFTMIN
    else:
        vol, mode = torch.stack(warped), L.COST_WARP_ONLY
    gen = torch.Generator().manual_seed(9)
    gvol = torch.randn(vol.shape, generator=gen)
    vol.backward(gvol)
    cl = lambda x: ops.to_channels_last(x.detach().cuda(), torch.float32)
    cams = ops.proj_cams_device(proj.cuda().contiguous(), 0)
    if mode == L.COST_WARP_ONLY:
        g_cl = torch.stack([ops.to_channels_last(gv.cuda(), torch.float32) for gv in gvol])
    else:
        g_cl = ops.to_channels_last(gvol.cuda(), torch.float32)
    dref, dsrcs, dtemp = ops.warp_cost_bwd(None if mode == L.COST_WARP_ONLY else cl(feats[0]), [cl(f) for f in feats[1:]], cams,
                                           dv.cuda(), g_cl, geom=L.GEOM_PROJ, cost=mode, temp=0.7, ref_hw=(h, w),
                                           want_dtemp=(mode == L.COST_SOFTMIN))
    for i in range(1, V):
        check_close(f"d src{i}", dsrcs[i - 1].permute(0, 3, 1, 2).cpu(), feats[i].grad, rel_l2=2e-4)
    if mode != L.COST_WARP_ONLY:
        check_close("d ref", dref.permute(0, 3, 1, 2).cpu(), feats[0].grad, rel_l2=2e-4)
    if mode == L.COST_SOFTMIN:
        check_close("d temp", dtemp.cpu(), temp.grad, rel_l2=1e-3)


def _train_step_engine(agg, H, W, V, D, seed, scene_seed, B, dtype):
    from wild_deep_mvs_amd import synthetic
    from wild_deep_mvs_amd.models.MVSNet.model import MVSNet
    net = MVSNet(agg)
    net.load_state_dict(synthetic.train_state_dict("mvsnet", synthetic.template_of(net), seed=seed))
    net = net.cuda().train()
    net.num_depth = D
    net.train_storage_dtype = dtype
    scene = synthetic.make_scene(B, V, H, W, seed=scene_seed)
    out = net(*[scene[k].cuda() for k in ("imgs", "K", "R", "t", "depth_min", "depth_max")])
    depth = out["depth"]
    assert depth.requires_grad and not out["photometric_confidence"].requires_grad
    gt, mask = synthetic.train_target(scene, depth.shape[1], depth.shape[2])
    loss = synthetic.supervised_loss(depth, gt.cuda(), mask.cuda(), scene["depth_min"].cuda(), scene["depth_max"].cuda())
    loss.backward()
    torch.cuda.synchronize()
    return net, depth.detach().cpu(), float(loss)


def _reg_net(arch):
    """(regulariser module on the GPU in train() mode, cost-volume channels) with the training fixture's weights."""
    from wild_deep_mvs_amd import synthetic
    if arch == "mvsnet":
        from wild_deep_mvs_amd.models.MVSNet.model import MVSNet
        net = MVSNet("variance")
        net.load_state_dict(synthetic.train_state_dict("mvsnet", synthetic.template_of(net), seed=3))
        return net.cuda().train().cost_regularization, 32
    from wild_deep_mvs_amd.models.CVP_MVSNet.frontend import Frontend
    net = Frontend()
    net.load_state_dict(synthetic.train_state_dict("cvp", synthetic.template_of(net), seed=3))
    return net.cuda().train().model.cost_reg_refine, 16


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("arch", ["mvsnet", "cvp"])
def test_regress_fn_blockwise_against_autograd(arch, dtype):
    """training.RegressFn (train()-mode U-Net + softmax regression, forward AND backward on the engine), checked block
    by block: the engine records every block's tensors (training.TRACE) and each block is replayed through ATen autograd
    on exactly those tensors -- forward (conv -> batch-statistics BN -> ReLU -> + skip), then backward from the recorded
    upstream gradient: d weight, d gamma, d beta, d input (including the gradient that arrived over a skip connection).
    A random-weight BatchNorm net amplifies a 1-ulp difference ~3x per layer (measured), so end-to-end comparisons only
    see storage noise; per-block comparisons on shared inputs are sharp and cover the whole wiring of the executor."""
    from wild_deep_mvs_amd import ops, synthetic, training as T
    reg, C = _reg_net(arch)
    gen = torch.Generator().manual_seed(5)
    B, D, h, w = 2, 16, 24, 32
    cost = (torch.rand(B, C, D, h, w, generator=gen) * 0.5).to(dtype)
    dv = torch.linspace(2.0, 6.0, D).view(1, D).repeat(B, 1).contiguous()
    gd = torch.randn(B, h, w, generator=gen)
    cost_cl = ops.to_channels_last(cost.cuda(), dtype).requires_grad_(True)
    blocks = reg.train_blocks()
    T.TRACE = {}
    try:
        depth, conf = T.RegressFn.apply(blocks, dv.cuda(), dtype, cost_cl, *T.RegressFn.block_params(blocks))
        (depth * gd.cuda()).sum().backward()
        torch.cuda.synchronize()
        tr = T.TRACE
    finally:
        T.TRACE = None
    bf = dtype == torch.bfloat16
    q = lambda x: x.to(dtype).float()
    for b in blocks[:-1]:
        r = tr[b.name]
        x = _cf(r["x"]).requires_grad_(True)
        wq = q(b.weight.detach().cpu()).requires_grad_(True)
        gamma = b.bn.weight.detach().cpu().clone().requires_grad_(True)
        beta = b.bn.bias.detach().cpu().clone().requires_grad_(True)
        if b.transposed:
            y = F.conv_transpose3d(x, wq, None, stride=b.stride, padding=1, output_padding=b.stride - 1)
        else:
            y = F.conv3d(x, wq, None, stride=b.stride, padding=1)
        # fp32 accumulation in another order (k-steps split over the waves, MFMA internal order) can flip a 16-bit rounding where
        # the exact value sits on a rounding boundary: on the tiny coarse levels one flip is already rel-L2 1e-4.  So: either the
        # norm bar, or at most 0.2 % of the values differ and none by more than one ulp of the format.
        got_y, ref_y = _cf(r["y"]), q(y.detach())
        s_ = check_close(f"{b.name} raw conv", got_y, ref_y)
        ulp1 = (2.0 ** -7 if bf else 2.0 ** -10) * ref_y.abs().clamp(min=2.0 ** -14)
        flips = ((got_y - ref_y).abs() > 0).float().mean()
        assert s_["rel_l2"] <= 1e-4 or (bool(((got_y - ref_y).abs() <= ulp1).all()) and float(flips) <= 2e-3), (s_, float(flips))
        # BN + ReLU (+ skip) on the ENGINE's stored y, so that the mask and the statistics are shared
        ye = _cf(r["y"]).requires_grad_(True)
        z = F.relu(F.batch_norm(ye, None, None, gamma, beta, training=True, eps=b.bn.eps))
        act = z if r["skip"] is None else z + _cf(r["skip"])
        check_close(f"{b.name} act", _cf(r["act"]), q(act.detach()), rel_l2=2e-4)
        dact = _cf(r["dact"])
        act.backward(dact)
        check_close(f"{b.name} d gamma", r["dgamma"].cpu(), gamma.grad, rel_l2=2e-4)
        check_close(f"{b.name} d beta", r["dbeta"].cpu(), beta.grad, rel_l2=2e-4)
        check_close(f"{b.name} dy", _cf(r["dy"]), q(ye.grad), rel_l2=2e-3 if bf else 3e-4)
        # conv backward from the ENGINE's stored dy
        y.backward(_cf(r["dy"]))
        check_close(f"{b.name} d weight", r["dw"].cpu(), wq.grad, rel_l2=1e-4)
        dx = x.grad if r["dx_prev"] is None else x.grad + _cf(r["dx_prev"])
        check_close(f"{b.name} d input", _cf(r["dx"]), q(dx), rel_l2=2e-3 if bf else 3e-4)
        if b.skip:   # the gradient handed to the skip source is the block's upstream gradient itself
            assert tr[b.skip]["dact"] is not None
    # head: 1-channel conv with bias -> softmax -> regression
    head = blocks[-1]
    r = tr[head.name]
    x = _cf(r["x"]).requires_grad_(True)
    wq = q(head.weight.detach().cpu()).requires_grad_(True)
    bias = head.conv_bias.detach().cpu().clone().requires_grad_(True)
    logits = F.conv3d(x, wq, bias, padding=1).squeeze(1)
    check_close("logits", r["logits"].cpu(), logits.detach(), rel_l2=1e-5)
    le = r["logits"].cpu().clone().requires_grad_(True)
    ((F.softmax(le, 1) * dv.view(B, D, 1, 1)).sum(1) * gd).sum().backward()
    check_close("d logits", r["dl8"][..., 0].float().cpu(), q(le.grad), rel_l2=2e-3 if bf else 3e-4)
    logits.backward(r["dl8"][..., 0].float().cpu())
    check_close("prob d weight", r["dw"].cpu(), wq.grad, rel_l2=1e-4)
    check_close("prob d bias", r["dbias"].cpu(), bias.grad, max_abs=1e-4 * float(wq.grad.abs().max()) + 1e-6)
    check_close("prob d input", _cf(r["dx"]), q(x.grad), rel_l2=2e-3 if bf else 3e-4)
    check_close("d cost (returned by autograd)", _cf(cost_cl.grad), _cf(tr[blocks[0].name]["dx"]), max_abs=0.0)
    for blk in blocks:   # every parameter received the recorded gradient
        assert torch.equal(blk.weight.grad, tr[blk.name]["dw"])


def _grad_report(tag, net, o_grads):
    """Per-tensor relative L2 (against the tensor's own norm, floored at 1e-3 of the largest gradient tensor: a gradient
    that is zero by symmetry -- prob.bias under the softmax -- has no relative error) and the cosine of the whole vector."""
    floor = 1e-3 * max(float(g.norm()) for g in o_grads.values())
    worst, rows, dot, n1, n2 = 0.0, [], 0.0, 0.0, 0.0
    for k, p in net.named_parameters():
        assert p.grad is not None, f"no gradient for {k}"
        got, ref = p.grad.detach().float().cpu(), o_grads[k].float()
        assert torch.isfinite(got).all(), k
        rel = float((got - ref).norm() / max(float(ref.norm()), floor))
        rows.append((k, rel))
        worst = max(worst, rel)
        dot += float((got * ref).sum()); n1 += float((got * got).sum()); n2 += float((ref * ref).sum())
    cos = dot / (n1 ** 0.5 * n2 ** 0.5)
    print(f"[train parity] {tag}: worst per-tensor rel-L2 {worst:.3e} ({max(rows, key=lambda r: r[1])[0]}), cosine {cos:.6f}")
    return worst, cos, rows


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("fname,agg", [("mvsnet_train.npz", "variance"), ("mvsnet_s_train.npz", "softmin")])
def test_mvsnet_train_step(fname, agg, dtype):
    """One full training step of the MVSNet mirror (forward in train() mode + loss.backward()).

    (1) Against the oracle's autograd with the engine's 16-bit storage emulated.  Unlike the U-Net-only test above the two
        sides do not see identical inputs here (the GPU 2-D extractor and the fp32 sample coordinates differ at the 1e-6
        level and the variance's cancellation lifts that to the storage ulp), so they are two realisations of the same
        storage noise: bounds sit at the noise level.
    (2) Against the plain fp32 oracle (pinned to the reference's own step by tests/test_oracle_train.py) and the
        reference golden: depth within the path's bar; gradients reported and bounded by the cosine of the full gradient
        vector -- at this fixture size (12 voxels per channel and image at the coarsest level, logits std ~2.7) the
        softmax Jacobian amplifies storage noise; the U-Net-only test shows that is precision, not arithmetic."""
    from test_oracle_train import oracle_train_step
    g = load_golden(fname)
    H, W, V, D, seed, scene_seed, B = [int(x) for x in g["meta"]]
    net, depth, loss = _train_step_engine(agg, H, W, V, D, seed, scene_seed, B, dtype)
    bf = dtype == torch.bfloat16
    # (1) storage-emulated oracle
    e_depth, e_loss, e_grads, e_stats = oracle_train_step(agg, H, W, V, D, seed, scene_seed, B, store=dtype)
    check_close("depth vs storage-emulated oracle", depth, e_depth, rel_l1=4e-3 if bf else 6e-4)
    worst, cos_e, rows_e = _grad_report(f"{agg} {dtype} vs storage-emulated oracle", net, e_grads)
    sd = net.state_dict()
    for k, ref in e_stats.items():
        check_close(f"stat {k}", sd[k].cpu(), ref, rel_l2=2e-2 if bf else 3e-3)
    assert int(sd["cost_regularization.conv0.bn.num_batches_tracked"]) == 2   # the fixture starts at 1
    # (2) fp32 oracle / reference golden
    o_depth, o_loss, o_grads, o_stats = oracle_train_step(agg, H, W, V, D, seed, scene_seed, B)
    # the yardstick is what the storage format itself costs: the storage-emulated oracle's own distance from the fp32 oracle
    # (depth: relative L1; gradient: 1 - cosine of the full vector).  The engine may exceed it by 15 % (depth) / 50 % (the
    # gradient: two independent realisations of the same rounding noise) -- a kernel regression cannot hide under a constant.
    emul_depth = float((e_depth - o_depth).abs().mean() / o_depth.abs().mean())
    dot = sum(float((e_grads[k].float() * o_grads[k].float()).sum()) for k in o_grads)
    n1 = sum(float((e_grads[k].float() ** 2).sum()) for k in o_grads) ** 0.5
    n2 = sum(float((o_grads[k].float() ** 2).sum()) for k in o_grads) ** 0.5
    emul_gap = 1.0 - dot / (n1 * n2)
    s = check_close("depth vs oracle", depth, o_depth)
    print(f"[train parity] {agg} {dtype}: depth rel-L1 engine {s['rel_l1']:.3e} / storage-emulated oracle {emul_depth:.3e}; "
          f"1 - cos(gradient) storage-emulated oracle {emul_gap:.3e}", flush=True)
    assert s["rel_l1"] <= 1.15 * emul_depth + 5e-5, (s, emul_depth)
    check_close("depth vs reference golden", depth, t(g["depth"]), rel_l1=1.15 * emul_depth + 2.5e-4)
    assert abs(loss - o_loss) <= (2e-2 if bf else 3e-3) * abs(o_loss), (loss, o_loss)
    worst, cos, rows = _grad_report(f"{agg} {dtype} vs fp32 oracle", net, o_grads)
    assert 1.0 - cos <= 1.5 * emul_gap + 2e-3, (cos, emul_gap, rows)
    assert 1.0 - cos_e <= 1.0 * emul_gap + 2e-3, (cos_e, emul_gap, rows_e)     # (1): engine vs the storage-emulated oracle itself


def test_train_step_then_eval_and_optimizer():
    """A few Adam steps through the engine lower the loss, and the model still runs in eval() afterwards (packed eval
    weights are rebuilt from the updated parameters)."""
    from wild_deep_mvs_amd import synthetic
    from wild_deep_mvs_amd.models.MVSNet.model import MVSNet
    net = MVSNet("variance")
    net.load_state_dict(synthetic.train_state_dict("mvsnet", synthetic.template_of(net), seed=0))
    net = net.cuda().train()
    net.num_depth = 16
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    scene = {k: v.cuda() for k, v in synthetic.make_scene(1, 3, 64, 96, seed=0).items() if isinstance(v, torch.Tensor)}
    gt, mask = synthetic.train_target({k: v.cpu() for k, v in scene.items()}, 16, 24)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        out = net(scene["imgs"], scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"])
        loss = synthetic.supervised_loss(out["depth"], gt.cuda(), mask.cuda(), scene["depth_min"], scene["depth_max"])
        loss.backward()
        opt.step()
        losses.append(float(loss))
    print("[train] losses", [f"{v:.3f}" for v in losses])
    assert losses[-1] < losses[0]
    net.eval()
    with torch.no_grad():
        out = net(scene["imgs"], scene["K"], scene["R"], scene["t"], scene["depth_min"], scene["depth_max"])
    assert torch.isfinite(out["depth"]).all()


@pytest.mark.parametrize("dtype", DT)
def test_cvp_train_step(dtype):
    """One training step of the CVP-MVSNet mirror in train() mode (48 coarse planes, halving refinement intervals, the
    regulariser applied per pyramid level, gradients through the bicubic depth upsampling) against the oracle's autograd
    and the reference's own step (tests/golden/cvp_train.npz; pinned by tests/test_oracle_train.py).  The per-block
    arithmetic is held tight by test_regress_fn_blockwise_against_autograd[cvp]; here depth per level, loss and the
    direction of the full gradient are checked (see test_mvsnet_train_step on why end-to-end gradients are loose)."""
    from test_oracle_train import cvp_oracle_train_step
    from wild_deep_mvs_amd import synthetic
    from wild_deep_mvs_amd.models.CVP_MVSNet.frontend import Frontend
    g = load_golden("cvp_train.npz")
    H, W, V, nscale, seed, scene_seed, bscale, B = [int(x) for x in g["meta"]]
    net = Frontend()
    net.load_state_dict(synthetic.train_state_dict("cvp", synthetic.template_of(net), seed=seed))
    net = net.cuda().train()
    net.train_storage_dtype = dtype
    scene = synthetic.make_scene(B, V, H, W, seed=scene_seed)
    scene["t"] = scene["t"] * bscale
    out = net(*[scene[k].cuda() for k in ("imgs", "K", "R", "t", "depth_min", "depth_max")], nscale=nscale)
    gt, mask = synthetic.train_target(scene, H, W)
    loss = synthetic.supervised_loss_list(out["depth_est_list"], gt.cuda(), mask.cuda(), scene["depth_min"].cuda(), scene["depth_max"].cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert tuple(out["photometric_confidence"].shape) == (B, 1, H, W) and not out["photometric_confidence"].requires_grad
    o_depths, o_loss, o_grads, o_stats = cvp_oracle_train_step(H, W, V, nscale, seed, scene_seed, bscale, B)
    bf = dtype == torch.bfloat16
    for i, d in enumerate(out["depth_est_list"]):
        check_close(f"depth_est_{i} vs oracle", d.detach().cpu(), o_depths[i], rel_l1=6e-3 if bf else 1e-3)
        check_close(f"depth_est_{i} vs reference golden", d.detach().cpu(), t(g[f"depth_est_{i}"]), rel_l1=6e-3 if bf else 1e-3)
    assert abs(float(loss) - o_loss) <= (3e-2 if bf else 5e-3) * abs(o_loss), (float(loss), o_loss)
    worst, cos, rows = _grad_report(f"cvp {dtype} vs fp32 oracle", net, o_grads)
    assert cos >= (0.9 if bf else 0.98), (cos, rows)
    sd = net.state_dict()
    for k, ref in o_stats.items():
        check_close(f"stat {k}", sd[k].cpu(), ref, rel_l2=3e-2 if bf else 5e-3)


# ---------------------------------------------------------------------------------------------------------------------
# Vis-MVSNet training pieces
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DT)
def test_vis_heads_and_fusion_backward(dtype):
    """pscv_softargmin_bwd with the expected-index and entropy heads, pscv_relu_bwd, the post-add ReLU of pscv_bn_act and
    pscv_fuse_pairs_bwd against ATen autograd on identical operands."""
    from wild_deep_mvs_amd import ops
    g = torch.Generator().manual_seed(21)
    B, D, h, w = 2, 12, 10, 14
    # softargmin heads (incl. a pixel whose softmax saturates so that clamp(p, 1e-9) is active)
    logits = (torch.randn(B, D, h, w, generator=g) * 2)
    logits[0, 3, 0, 0] = 60.0
    logits.requires_grad_(True)
    p = F.softmax(logits, 1)
    idx = (p * torch.arange(D).view(1, D, 1, 1)).sum(1)
    ent = (-p * p.clamp(1e-9, 1.0).log()).sum(1)
    gi, ge = torch.randn(B, h, w, generator=g), torch.randn(B, h, w, generator=g)
    (idx * gi + ent * ge).sum().backward()
    got = ops.softargmin_bwd(logits.detach().cuda(), None, None, torch.float16, grad_index=gi.cuda(), grad_entropy=ge.cuda())
    check_close("d logits (index + entropy)", got[..., 0].float().cpu(), logits.grad, rel_l2=1e-3)
    # relu after the add, forward and backward
    y = (torch.randn(B, 8, D, h, w, generator=g)).to(dtype).float()
    skip = torch.randn(B, 8, D, h, w, generator=g).to(dtype).float()
    sc, bi = torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g) * 0.2
    out_ref = F.relu(y * sc.view(1, 8, 1, 1, 1) + bi.view(1, 8, 1, 1, 1) + skip)
    out = ops.bn_act(_cl(y, dtype), sc.cuda(), bi.cuda(), relu="post", skip=_cl(skip, dtype))
    check_close("bn_act post relu", _cf(out), out_ref.to(dtype).float(), rel_l2=4e-3 if dtype == torch.bfloat16 else 5e-4)
    dout = torch.randn(B, 8, D, h, w, generator=g).to(dtype).float()
    dpre = ops.relu_bwd(_cl(dout, dtype), out)
    check_close("relu_bwd", _cf(dpre), dout * (_cf(out) > 0).float(), max_abs=0.0)
    # fusion
    n = 3
    Is = [torch.randn(B, 8, D, h, w, generator=g).to(dtype).float().requires_grad_(True) for _ in range(n)]
    us = [(torch.randn(B, h, w, generator=g) * 0.7).requires_grad_(True) for _ in range(n)]
    ws = [(-u).exp().view(B, 1, 1, h, w) for u in us]
    fused = sum(I * w_ for I, w_ in zip(Is, ws)) / sum(ws)
    G = torch.randn(B, 8, D, h, w, generator=g).to(dtype).float()
    fused.backward(G)
    dI, dU = ops.fuse_pairs_bwd([_cl(I.detach(), dtype) for I in Is], [u.detach().cuda() for u in us], _cl(G, dtype))
    for v in range(n):
        check_close(f"d interm {v}", _cf(dI[v]), Is[v].grad, rel_l2=4e-3 if dtype == torch.bfloat16 else 5e-4)
        check_close(f"d uncert {v}", dU[v].cpu(), us[v].grad, rel_l2=1e-4)


def _cg(x, w, dy, *, stride=1, transposed=False, pad=1):
    """(input gradient, weight gradient) of one (transposed) convolution from ATen autograd."""
    x = x.clone().requires_grad_(True)
    w = w.clone().requires_grad_(True)
    y = F.conv_transpose3d(x, w, None, stride=stride, padding=pad, output_padding=stride - 1) if transposed else \
        F.conv3d(x, w, None, stride=stride, padding=pad)
    y.backward(dy)
    return x.grad, w.grad


def _bn_bwd_ref(y, gamma, beta, dz_src, relu, eps=1e-5):
    y = y.clone().requires_grad_(True)
    gamma, beta = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    z = F.batch_norm(y, None, None, gamma, beta, training=True, eps=eps)
    (F.relu(z) if relu else z).backward(dz_src)
    return y.grad, gamma.grad, beta.grad


@pytest.mark.parametrize("dtype", DT)
def test_vis_unet_fn_stepwise_against_autograd(dtype):
    """training.VisUNetFn (the residual-block U-Net of Vis-MVSNet's Reg / RegFuse in train() mode, forward AND backward on
    the engine) step by step on the tensors the engine recorded (training.TRACE): every convolution, batch-statistics
    BatchNorm, post-add ReLU, the strided 1x1x1 shortcut, the linear decoder with its channel concatenation, and in the
    backward every d weight / d gamma / d beta / d input incl. the three-way gradient sum at the first block's output."""
    from wild_deep_mvs_amd import ops, synthetic, training as T
    from wild_deep_mvs_amd.models.VisMVSNet.frontend import Frontend
    net = Frontend()
    net.load_state_dict(synthetic.sharpened_state_dict("vis", synthetic.template_of(net), seed=2))
    holder = net.cuda().train().model.stage1.reg
    gen = torch.Generator().manual_seed(6)
    n, d, h, w = 2, 8, 12, 20
    x = torch.randn(n, 8, d, h, w, generator=gen).to(dtype)
    gout = torch.randn(n, 8, d, h, w, generator=gen).to(dtype)
    x_cl = ops.to_channels_last(x.cuda(), dtype).requires_grad_(True)
    params = T.VisUNetFn.params(holder)
    T.TRACE = {}
    try:
        out = T.VisUNetFn.apply(holder, dtype, 1, x_cl, *params)
        out.backward(ops.to_channels_last(gout.cuda(), dtype))
        torch.cuda.synchronize()
        f, b = T.TRACE["vis_unet"][0], T.TRACE["vis_unet_bwd"][0]
    finally:
        T.TRACE = None
    bf = dtype == torch.bfloat16
    q = lambda v: v.to(dtype).float()
    tol, btol = 2e-4, (2e-3 if bf else 3e-4)
    b0, b1, dec = T.VisUNetFn.parts(holder)
    W = lambda p_: q(p_.detach().cpu())
    P = lambda p_: p_.detach().cpu().float()
    R = {k: (_cf(v) if torch.is_tensor(v) and v.dim() == 5 else v) for k, v in {**f, **{"b_" + k: v for k, v in b.items() if v is not None}}.items()}
    bn = lambda y, m: F.batch_norm(y, None, None, P(m.weight), P(m.bias), training=True, eps=m.eps)
    # ---- forward ----
    check_close("y1", R["y1"], q(F.conv3d(R["x"], W(b0.conv1.weight), padding=1)), rel_l2=tol)
    check_close("t", R["t"], q(F.relu(bn(R["y1"], b0.bn1))), rel_l2=tol)
    check_close("y2", R["y2"], q(F.conv3d(R["t"], W(b0.conv2.weight), padding=1)), rel_l2=tol)
    check_close("enc0", R["enc0"], q(F.relu(bn(R["y2"], b0.bn2) + R["x"])), rel_l2=tol)
    check_close("y3", R["y3"], q(F.conv3d(R["enc0"], W(b1.conv1.weight), stride=2, padding=1)), rel_l2=tol)
    check_close("t1", R["t1"], q(F.relu(bn(R["y3"], b1.bn1))), rel_l2=tol)
    check_close("y4 (1x1x1 stride-2 shortcut)", R["y4"], q(F.conv3d(R["enc0"], W(b1.downsample[0].weight), stride=2, padding=0)), rel_l2=tol)
    check_close("ds", R["ds"], q(bn(R["y4"], b1.downsample[1])), rel_l2=tol)
    check_close("y5", R["y5"], q(F.conv3d(R["t1"], W(b1.conv2.weight), padding=1)), rel_l2=tol)
    check_close("e1", R["e1"], q(F.relu(bn(R["y5"], b1.bn2) + R["ds"])), rel_l2=tol)
    check_close("up", R["up"], q(F.conv_transpose3d(R["e1"], W(dec[0].weight), stride=2, padding=1, output_padding=1)), rel_l2=tol)
    check_close("cat", R["cat"], torch.cat([R["up"], R["enc0"]], 1), max_abs=0.0)
    check_close("out", R["out"], q(F.conv3d(R["cat"], W(dec[1].weight), padding=1)), rel_l2=tol)
    # ---- backward ----
    G = {k: p_.grad.detach().cpu().float() for k, p_ in zip(
        ["c1a", "g1", "b1", "c2a", "g2", "b2", "c1b", "g3", "b3", "ds", "g4", "b4", "c2b", "g5", "b5", "dec", "post"], params)}
    dx_, dw_ = _cg(R["cat"], W(dec[1].weight), R["b_g"])
    check_close("d cat", R["b_dcat"], q(dx_), rel_l2=btol); check_close("dW post", G["post"], dw_, rel_l2=tol)
    dx_, dw_ = _cg(R["e1"], W(dec[0].weight), R["b_dcat"][:, :8], stride=2, transposed=True)
    check_close("d e1", R["b_de1"], q(dx_), rel_l2=btol); check_close("dW deconv", G["dec"], dw_, rel_l2=tol)
    check_close("d pre (post-add relu)", R["b_dpre"], R["b_de1"] * (R["e1"] > 0).float(), max_abs=0.0)
    dy_, dg_, db_ = _bn_bwd_ref(R["y5"], P(b1.bn2.weight), P(b1.bn2.bias), R["b_dpre"], False, b1.bn2.eps)
    check_close("dy5", R["b_dy5"], q(dy_), rel_l2=btol); check_close("d gamma5", G["g5"], dg_, rel_l2=tol); check_close("d beta5", G["b5"], db_, rel_l2=tol)
    dx_, dw_ = _cg(R["t1"], W(b1.conv2.weight), R["b_dy5"])
    check_close("d t1", R["b_dt1"], q(dx_), rel_l2=btol); check_close("dW conv2 (block 1)", G["c2b"], dw_, rel_l2=tol)
    dy_, dg_, db_ = _bn_bwd_ref(R["y4"], P(b1.downsample[1].weight), P(b1.downsample[1].bias), R["b_dpre"], False, b1.downsample[1].eps)
    check_close("dy4", R["b_dy4"], q(dy_), rel_l2=btol); check_close("d gamma4", G["g4"], dg_, rel_l2=tol); check_close("d beta4", G["b4"], db_, rel_l2=tol)
    dx4, dw_ = _cg(R["enc0"], W(b1.downsample[0].weight), R["b_dy4"], stride=2, pad=0)
    check_close("dW shortcut", G["ds"], dw_, rel_l2=tol)
    dy_, dg_, db_ = _bn_bwd_ref(R["y3"], P(b1.bn1.weight), P(b1.bn1.bias), R["b_dt1"], True, b1.bn1.eps)
    check_close("dy3", R["b_dy3"], q(dy_), rel_l2=btol); check_close("d gamma3", G["g3"], dg_, rel_l2=tol); check_close("d beta3", G["b3"], db_, rel_l2=tol)
    dx3, dw_ = _cg(R["enc0"], W(b1.conv1.weight), R["b_dy3"], stride=2)
    check_close("dW conv1 (block 1)", G["c1b"], dw_, rel_l2=tol)
    check_close("d enc0 (concat slice + shortcut + strided conv)", R["b_denc0"], q(q(R["b_dcat"][:, 8:] + dx4) + dx3), rel_l2=btol)
    check_close("d pre0", R["b_dpre0"], R["b_denc0"] * (R["enc0"] > 0).float(), max_abs=0.0)
    dy_, dg_, db_ = _bn_bwd_ref(R["y2"], P(b0.bn2.weight), P(b0.bn2.bias), R["b_dpre0"], False, b0.bn2.eps)
    check_close("dy2", R["b_dy2"], q(dy_), rel_l2=btol); check_close("d gamma2", G["g2"], dg_, rel_l2=tol); check_close("d beta2", G["b2"], db_, rel_l2=tol)
    dx_, dw_ = _cg(R["t"], W(b0.conv2.weight), R["b_dy2"])
    check_close("d t", R["b_dt"], q(dx_), rel_l2=btol); check_close("dW conv2 (block 0)", G["c2a"], dw_, rel_l2=tol)
    dy_, dg_, db_ = _bn_bwd_ref(R["y1"], P(b0.bn1.weight), P(b0.bn1.bias), R["b_dt"], True, b0.bn1.eps)
    check_close("dy1", R["b_dy1"], q(dy_), rel_l2=btol); check_close("d gamma1", G["g1"], dg_, rel_l2=tol); check_close("d beta1", G["b1"], db_, rel_l2=tol)
    dx_, dw_ = _cg(R["x"], W(b0.conv1.weight), R["b_dy1"])
    check_close("dW conv1 (block 0)", G["c1a"], dw_, rel_l2=tol)
    check_close("d x (conv path + residual)", R["b_dx"], q(dx_ + R["b_dpre0"]), rel_l2=btol)
    check_close("d x returned by autograd", _cf(x_cl.grad), R["b_dx"], max_abs=0.0)


def test_frozen_batchnorm_submodule_in_a_training_step():
    """A BatchNorm submodule put in eval() inside a train()-mode network (frozen-BN fine-tuning) is honoured per module like
    nn.BatchNorm: it normalises with its running statistics, they and num_batches_tracked stay untouched, and its backward is
    the eval-mode one (dy = gamma * invstd * dz; d gamma / d beta against the running statistics) -- while its train()-mode
    sibling in the same block keeps using and updating batch statistics."""
    from wild_deep_mvs_amd import ops, synthetic, training as T
    from wild_deep_mvs_amd.models.VisMVSNet.frontend import Frontend
    dtype = torch.float16
    net = Frontend()
    net.load_state_dict(synthetic.sharpened_state_dict("vis", synthetic.template_of(net), seed=2))
    holder = net.cuda().train().model.stage1.reg
    b0, b1, dec = T.VisUNetFn.parts(holder)
    gen = torch.Generator().manual_seed(9)
    with torch.no_grad():
        b0.bn1.running_mean.copy_(0.1 * torch.randn(8, generator=gen))
        b0.bn1.running_var.copy_(0.5 + torch.rand(8, generator=gen))
    b0.bn1.eval()
    rm0, rv0, nb0 = b0.bn1.running_mean.clone(), b0.bn1.running_var.clone(), int(b0.bn1.num_batches_tracked)
    nb2 = int(b0.bn2.num_batches_tracked)
    n, d, h, w = 2, 8, 12, 20
    x = torch.randn(n, 8, d, h, w, generator=gen).to(dtype)
    gout = torch.randn(n, 8, d, h, w, generator=gen).to(dtype)
    x_cl = ops.to_channels_last(x.cuda(), dtype).requires_grad_(True)
    params = T.VisUNetFn.params(holder)
    T.TRACE = {}
    try:
        out = T.VisUNetFn.apply(holder, dtype, 1, x_cl, *params)
        out.backward(ops.to_channels_last(gout.cuda(), dtype))
        torch.cuda.synchronize()
        f, b = T.TRACE["vis_unet"][0], T.TRACE["vis_unet_bwd"][0]
    finally:
        T.TRACE = None
    assert torch.equal(b0.bn1.running_mean, rm0) and torch.equal(b0.bn1.running_var, rv0) and int(b0.bn1.num_batches_tracked) == nb0
    assert int(b0.bn2.num_batches_tracked) == nb2 + 1
    q = lambda v: v.to(dtype).float()
    P = lambda p_: p_.detach().cpu().float()
    y1, t_, dt = _cf(f["y1"]), _cf(f["t"]), _cf(b["dt"])
    y = y1.clone().requires_grad_(True)
    gamma, beta = P(b0.bn1.weight).requires_grad_(True), P(b0.bn1.bias).requires_grad_(True)
    z = F.relu(F.batch_norm(y, rm0.cpu(), rv0.cpu(), gamma, beta, training=False, eps=b0.bn1.eps))
    check_close("frozen BN forward", t_, q(z.detach()), rel_l2=2e-4)
    z.backward(dt)
    check_close("frozen BN dy", _cf(b["dy1"]), q(y.grad), rel_l2=3e-4)
    check_close("frozen BN d gamma", b0.bn1.weight.grad.cpu().float(), gamma.grad, rel_l2=2e-4)
    check_close("frozen BN d beta", b0.bn1.bias.grad.cpu().float(), beta.grad, rel_l2=2e-4)


@pytest.mark.parametrize("dtype", DT)
def test_vis_train_step(dtype):
    """One training step of the Vis-MVSNet mirror in train() mode (three detached cascade stages; per source view: fused warp
    + group correlation, pair U-Net, score head with expected index + entropy, 2-D uncertainty net; visibility-weighted
    fusion; fuse U-Net) with the reference trainer's loss (fused L1 + Bayesian pair loss) against the oracle's autograd and
    the reference's own step (tests/golden/vis_train.npz; pinned by tests/test_oracle_train.py)."""
    from test_oracle_train import vis_oracle_train_step
    from wild_deep_mvs_amd import synthetic
    from wild_deep_mvs_amd.models.VisMVSNet.frontend import Frontend
    g = load_golden("vis_train.npz")
    H, W, V, seed, scene_seed, B = [int(x) for x in g["meta"]]
    depth_nums, scales = [int(x) for x in g["depth_nums"]], [float(x) for x in g["interval_scales"]]
    net = Frontend()
    net.load_state_dict(synthetic.sharpened_state_dict("vis", synthetic.template_of(net), seed=seed))
    net = net.cuda().train()
    net.depth_nums, net.interval_scales = depth_nums, scales
    net.train_storage_dtype = dtype
    scene = synthetic.make_scene(B, V, H, W, seed=scene_seed)
    out = net(*[scene[k].cuda() for k in ("imgs", "K", "R", "t", "depth_min", "depth_max")])
    gt, mask = synthetic.train_target(scene, H // 2, W // 2)
    loss = synthetic.vis_supervised_loss(out, gt.cuda(), mask.cuda(), scene["depth_min"].cuda(), scene["depth_max"].cuda(), V)
    loss.backward()
    torch.cuda.synchronize()
    assert tuple(out["photometric_confidence"].shape) == (B, 3, H // 2, W // 2)
    o_out, o_loss, o_grads, o_stats = vis_oracle_train_step(H, W, V, seed, scene_seed, B, tuple(depth_nums), tuple(scales))
    bf = dtype == torch.bfloat16
    for i, d in enumerate(out["depth_est_list"]):
        check_close(f"depth_est_{i} vs oracle", d.detach().cpu(), o_out["depth_est_list"][i].detach(), rel_l1=6e-3 if bf else 1e-3)
        check_close(f"depth_est_{i} vs reference golden", d.detach().cpu(), t(g[f"depth_est_{i}"]), rel_l1=6e-3 if bf else 1e-3)
    for i, prs in enumerate(out["depth_pair_list"]):
        for j, (dp, (unc,)) in enumerate(prs):
            check_close(f"pair {i},{j} depth", dp.detach().cpu(), t(g[f"pair_{i}_{j}_depth"]), rel_l1=6e-3 if bf else 1e-3)
            check_close(f"pair {i},{j} uncertainty", unc.detach().cpu(), t(g[f"pair_{i}_{j}_uncert"]), rel_l2=8e-2 if bf else 1.5e-2)
    assert abs(float(loss) - o_loss) <= (3e-2 if bf else 5e-3) * abs(o_loss), (float(loss), o_loss)
    worst, cos, rows = _grad_report(f"vis {dtype} vs fp32 oracle", net, o_grads)
    assert cos >= (0.9 if bf else 0.98), (cos, rows)
    sd = net.state_dict()
    for k, ref in o_stats.items():
        check_close(f"stat {k}", sd[k].cpu(), ref, rel_l2=5e-2 if bf else 8e-3)


def test_function_level_warps_are_differentiable():
    """models.MVSNet.module.homo_warping, models.CVP_MVSNet.models.modules.homo_warping and
    models.VisMVSNet.homography.homography_warping carry autograd to the source map (training.WarpOnlyFn), like the
    reference's grid_sample-based functions: gradients against ATen autograd through the oracle's versions."""
    from oracle import mvsnet as O, cvpmvsnet as OC, vismvsnet as OV
    from wild_deep_mvs_amd.models.MVSNet.module import homo_warping as hw_mvs
    from wild_deep_mvs_amd.models.CVP_MVSNet.models.modules import homo_warping as hw_cvp
    from wild_deep_mvs_amd.models.VisMVSNet.homography import homography_warping as hw_vis
    proj, dv, feats = _sweep_case(32, 3, 16, 24, 8, seed=4)
    gen = torch.Generator().manual_seed(1)
    # MVSNet
    src = feats[1].clone().requires_grad_(True)
    ref = O.homo_warping(src, proj[:, 1], proj[:, 0], dv, (16, 24))
    gv = torch.randn(ref.shape, generator=gen)
    ref.backward(gv)
    s2 = feats[1].clone().cuda().requires_grad_(True)
    got = hw_mvs(s2, proj[:, 1].cuda(), proj[:, 0].cuda(), dv.cuda(), (16, 24))
    got.backward(gv.cuda())
    check_close("mvsnet homo_warping", got.detach().cpu(), ref.detach(), max_abs=3e-4)
    check_close("mvsnet homo_warping d src", s2.grad.cpu(), src.grad, rel_l2=2e-4)
    # CVP (16 channels, per-pixel hypotheses)
    from wild_deep_mvs_amd import synthetic
    scene = synthetic.make_scene(1, 2, 64, 96, seed=4)
    row = torch.tensor([0., 0., 0., 1.])
    ex = [torch.cat((torch.cat((scene["R"][:, i], scene["t"][:, i] * 8), 2), row.view(1, 1, 4)), 1) for i in range(2)]
    K = scene["K"].clone()
    K[:, :, :2] /= 4
    hyp = 3.0 + torch.rand(1, 4, 16, 24, generator=gen)
    src = (torch.randn(1, 16, 16, 24, generator=gen) * 0.5).requires_grad_(True)
    ref = OC.homo_warping(src, K[:, 0], K[:, 1], ex[0], ex[1], hyp, (16, 24))
    gv = torch.randn(ref.shape, generator=gen)
    ref.backward(gv)
    s2 = src.detach().clone().cuda().requires_grad_(True)
    got = hw_cvp(s2, K[:, 0].cuda(), K[:, 1].cuda(), ex[0].cuda(), ex[1].cuda(), hyp.cuda(), (16, 24))
    got.backward(gv.cuda())
    check_close("cvp homo_warping", got.detach().cpu(), ref.detach(), max_abs=3e-4)
    check_close("cvp homo_warping d src", s2.grad.cpu(), src.grad, rel_l2=2e-4)
    # Vis (one homography per batch item)
    H = torch.eye(3).view(1, 3, 3) + 0.02 * torch.randn(1, 3, 3, generator=gen)
    src = (torch.randn(1, 32, 16, 24, generator=gen) * 0.5).requires_grad_(True)
    ref = OV.homography_warping(src, H.view(1, 1, 1, 3, 3), (16, 24))
    gv = torch.randn(ref.shape, generator=gen)
    ref.backward(gv)
    s2 = src.detach().clone().cuda().requires_grad_(True)
    got = hw_vis(s2, H.cuda(), (16, 24))
    got.backward(gv.cuda())
    check_close("vis homography_warping", got.detach().cpu(), ref.detach(), max_abs=3e-4)
    check_close("vis homography_warping d src", s2.grad.cpu(), src.grad, rel_l2=2e-4)


@pytest.mark.parametrize("dtype", DT)
def test_conv2d_k5s2_gradients(dtype):
    """The two decompositions FeatureNetFn uses for a k5 s2 p2 Conv2d: weight gradient from the four parity planes of the
    input (four single-plane runs of the 3-D MFMA weight-gradient kernel) and data gradient as four parity sub-convolutions
    on the forward conv2d kernel; plus the k3 s1 forms.  Against ATen autograd on identical operands."""
    from wild_deep_mvs_amd import ops, training as T
    g = torch.Generator().manual_seed(31)
    B, H, W = 2, 24, 40
    cl2 = lambda v: v.permute(0, 2, 3, 1).to(dtype).contiguous().cuda()
    cf2 = lambda v: v.float().permute(0, 3, 1, 2).cpu()
    for ci, co in ((8, 16), (16, 32)):
        w = (torch.randn(co, ci, 5, 5, generator=g) * 0.1).to(dtype).float().requires_grad_(True)
        x = torch.randn(B, ci, H, W, generator=g).to(dtype).float().requires_grad_(True)
        y = F.conv2d(x, w, None, stride=2, padding=2)
        dy = torch.randn(y.shape, generator=g).to(dtype).float()
        y.backward(dy)
        dw = T._wgrad2d_k5s2(cl2(dy), cl2(x.detach()), co, ci)
        check_close(f"k5s2 dW {ci}->{co}", dw.cpu(), w.grad, rel_l2=2e-5)
        dx = torch.empty((B, H, W, ci), dtype=torch.float32, device="cuda")
        for par, sub in enumerate(T._dgrad2d_k5s2_layers(w.detach().cuda(), dtype)):
            ops.conv2d(cl2(dy), sub, out=dx, parity=par, out_dtype=torch.float32)
        check_close(f"k5s2 dX {ci}->{co}", cf2(dx), x.grad, rel_l2=2e-5)
    for ci, co in ((8, 8), (16, 16), (32, 32)):
        w = (torch.randn(co, ci, 3, 3, generator=g) * 0.1).to(dtype).float().requires_grad_(True)
        x = torch.randn(B, ci, H, W, generator=g).to(dtype).float().requires_grad_(True)
        y = F.conv2d(x, w, None, padding=1)
        dy = torch.randn(y.shape, generator=g).to(dtype).float()
        y.backward(dy)
        check_close(f"k3 dW {ci}->{co}", T._wgrad2d_k3(cl2(dy), cl2(x.detach()), co, ci).cpu(), w.grad, rel_l2=2e-5)
        dx = ops.conv2d(cl2(dy), T._dgrad2d_k3_layer(w.detach().cuda(), dtype), out_dtype=torch.float32)
        check_close(f"k3 dX {ci}->{co}", cf2(dx), x.grad, rel_l2=2e-5)


@pytest.mark.parametrize("dtype", DT)
def test_feature_net_fn_against_module_autograd(dtype):
    """training.FeatureNetFn (MVSNet's 2-D extractor in train() mode, forward and backward on the engine) against the same
    module under PyTorch-ROCm autograd (fp32): features, every parameter gradient (direction of the whole vector + per-tensor
    errors reported) and the BatchNorm2d running statistics.  Eight stored 16-bit layers each way; the kernels themselves are
    held tight by test_conv2d_k5s2_gradients / test_bn_backward / test_conv3d_wgrad_and_dgrad."""
    from wild_deep_mvs_amd import synthetic, training as T
    from wild_deep_mvs_amd.models.MVSNet.model import MVSNet
    import copy
    net = MVSNet("variance")
    net.load_state_dict(synthetic.train_state_dict("mvsnet", synthetic.template_of(net), seed=0))
    fa = net.feature.cuda().train()
    fb = copy.deepcopy(fa)
    gen = torch.Generator().manual_seed(4)
    img = torch.rand(2, 3, 64, 96, generator=gen).cuda()
    gout = torch.randn(2, 32, 16, 24, generator=gen).cuda()
    out = T.FeatureNetFn.apply(fa, dtype, 1, img, *T.FeatureNetFn.params(fa))
    out.backward(gout.permute(0, 2, 3, 1).to(dtype).contiguous())
    ref = fb(img)
    ref.backward(gout)
    torch.cuda.synchronize()
    bf = dtype == torch.bfloat16
    check_close("features", out.detach().float().permute(0, 3, 1, 2).cpu(), ref.detach().cpu(), rel_l2=3e-2 if bf else 4e-3)
    o_grads = {k: p.grad.detach().cpu() for k, p in fb.named_parameters()}
    worst, cos, rows = _grad_report(f"FeatureNetFn {dtype} vs module autograd", fa, o_grads)
    assert cos >= (0.97 if bf else 0.995), (cos, rows)
    for (k, a), (_, b_) in zip(fa.state_dict().items(), fb.state_dict().items()):
        if "running_" in k:
            check_close(f"stat {k}", a.cpu(), b_.cpu(), rel_l2=3e-2 if bf else 4e-3)


@pytest.mark.parametrize("dtype", DT)
def test_feature_net_fn_grouped_equals_per_view_calls(dtype):
    """All views of a sample in ONE FeatureNetFn pass (groups = views: pscv_bn_*_grouped, every view normalised with its own batch
    statistics) against one pass per view on a copy of the module, the way the reference drives its extractor
    (models/MVSNet/model.py:101-107): the same features per view bit for bit (same kernels per image, per-group statistics summed in
    the same order), the running statistics after the V sequential updates, and every parameter gradient (sums over the views:
    fp32 summation order differs, 1e-5)."""
    from wild_deep_mvs_amd import synthetic, training as T
    from wild_deep_mvs_amd.models.MVSNet.model import MVSNet
    import copy
    net = MVSNet("variance")
    net.load_state_dict(synthetic.train_state_dict("mvsnet", synthetic.template_of(net), seed=0))
    fa = net.feature.cuda().train()
    fb = copy.deepcopy(fa)
    nbt0 = int(fa.conv0.bn.num_batches_tracked)
    gen = torch.Generator().manual_seed(9)
    V, B = 3, 2
    imgs = [torch.rand(B, 3, 40, 72, generator=gen).cuda() for _ in range(V)]
    gouts = [torch.randn(B, 10, 18, 32, generator=gen).to(dtype).cuda() for _ in range(V)]
    out = T.FeatureNetFn.apply(fa, dtype, V, torch.cat(imgs, 0), *T.FeatureNetFn.params(fa))
    out.backward(torch.cat(gouts, 0))
    refs = []
    for img, go in zip(imgs, gouts):
        o = T.FeatureNetFn.apply(fb, dtype, 1, img, *T.FeatureNetFn.params(fb))
        o.backward(go)
        refs.append(o.detach())
    torch.cuda.synchronize()
    assert torch.equal(out.detach(), torch.cat(refs, 0)), "grouped features differ from the per-view passes"
    for (k, a), (_, b_) in zip(fa.state_dict().items(), fb.state_dict().items()):
        if "running_" in k:
            check_close(f"stat {k}", a.float().cpu(), b_.float().cpu(), rel_l2=1e-6)
        if "num_batches_tracked" in k:
            assert int(a) == int(b_) == nbt0 + V
    for (k, pa), (_, pb) in zip(fa.named_parameters(), fb.named_parameters()):
        check_close(f"grad {k}", pa.grad.float().cpu(), pb.grad.float().cpu(), rel_l2=2e-5)


@pytest.mark.parametrize("dtype", DT)
def test_vis_unet_fn_grouped_equals_per_view_calls(dtype):
    """The pair U-Net of ALL source views in one VisUNetFn pass (groups = views: every view normalised with its own batch
    statistics) against one pass per view on a copy of the module, the way the reference runs its pair branch
    (models/VisMVSNet/model_cas.py:341-352): outputs bit for bit, running statistics, parameter gradients (sums over the views)."""
    from wild_deep_mvs_amd import synthetic, training as T
    from wild_deep_mvs_amd.models.VisMVSNet.frontend import Frontend
    import copy
    net = Frontend()
    net.load_state_dict(synthetic.train_state_dict("vis", synthetic.template_of(net), seed=0))
    ha = [m for m in net.modules() if hasattr(m, "reg") and hasattr(m, "reg_fuse")][0].reg.cuda().train()
    hb = copy.deepcopy(ha)
    gen = torch.Generator().manual_seed(12)
    V, n, d, h, w = 3, 2, 8, 12, 20
    xs = [(torch.randn(n, d, h, w, 8, generator=gen) * 0.5).to(dtype).cuda() for _ in range(V)]
    gs = [(torch.randn(n, d, h, w, 8, generator=gen) * 0.1).to(dtype).cuda() for _ in range(V)]
    xa = torch.cat(xs, 0).requires_grad_(True)
    out = T.VisUNetFn.apply(ha, dtype, V, xa, *T.VisUNetFn.params(ha))
    out.backward(torch.cat(gs, 0))
    refs, dxs = [], []
    for x, g in zip(xs, gs):
        xr = x.clone().requires_grad_(True)
        o = T.VisUNetFn.apply(hb, dtype, 1, xr, *T.VisUNetFn.params(hb))
        o.backward(g)
        refs.append(o.detach()); dxs.append(xr.grad)
    torch.cuda.synchronize()
    assert torch.equal(out.detach(), torch.cat(refs, 0)), "grouped pair U-Net differs from the per-view passes"
    assert torch.equal(xa.grad, torch.cat(dxs, 0)), "grouped input gradient differs from the per-view passes"
    for (k, a), (_, b_) in zip(ha.state_dict().items(), hb.state_dict().items()):
        if "running_" in k:
            check_close(f"stat {k}", a.float().cpu(), b_.float().cpu(), rel_l2=1e-6)
    for (k, pa), (_, pb) in zip(ha.named_parameters(), hb.named_parameters()):
        if pb.grad is not None:
            check_close(f"grad {k}", pa.grad.float().cpu(), pb.grad.float().cpu(), rel_l2=3e-5)


@pytest.mark.parametrize("dtype", DT)
def test_generic_2d_layer_nodes_against_aten_autograd(dtype):
    """training.Conv2dFn (k3 s1 | k3 s2 | k1 s1 | k1 s2, up to 128 channels: weight gradients in 64-channel slices, stride-2 gradients
    from parity planes / parity sub-convolutions), Deconv2dFn (ConvTranspose2d k3 s2 p1 op1) and BnAct2dFn (grouped batch-statistics
    BatchNorm2d, ReLU before / after a skip add) against ATen autograd on identical 16-bit-exact operands."""
    from wild_deep_mvs_amd import ops, training as T
    g = torch.Generator().manual_seed(77)
    cl = lambda v: v.permute(0, 2, 3, 1).to(dtype).contiguous().cuda()
    cf = lambda v: v.float().permute(0, 3, 1, 2).cpu()
    holder = torch.nn.Module()
    N, H, W = 2, 16, 24
    for tag, (ci, co, k, s_) in enumerate([(32, 32, 3, 1), (64, 128, 3, 2), (128, 128, 3, 1), (32, 64, 3, 2), (16, 32, 1, 1), (64, 128, 1, 2),
                                           (128, 64, 3, 1), (128, 32, 3, 1)]):
        w = (torch.randn(co, ci, k, k, generator=g) * 0.1).to(dtype).float().requires_grad_(True)
        x = torch.randn(N, ci, H, W, generator=g).to(dtype).float().requires_grad_(True)
        y = F.conv2d(x, w, None, stride=s_, padding=k // 2)
        dy = (torch.randn(y.shape, generator=g) * 0.5).to(dtype).float()
        y.backward(dy)
        xe = cl(x.detach()).requires_grad_(True)
        we = w.detach().cuda().requires_grad_(True)
        ye = T.Conv2dFn.apply(holder, f"t{tag}", dtype, s_, xe, we)
        ye.backward(cl(dy))
        ulp = 2 ** -8 if dtype == torch.bfloat16 else 2 ** -11
        check_close(f"conv k{k}s{s_} {ci}->{co} y", cf(ye.detach()), y.detach(), rel_l2=2 * ulp)
        check_close(f"conv k{k}s{s_} {ci}->{co} dW", we.grad.cpu(), w.grad, rel_l2=3e-5)
        check_close(f"conv k{k}s{s_} {ci}->{co} dX", cf(xe.grad), x.grad, rel_l2=2 * ulp)
    for tag, (ci, co) in enumerate([(128, 64), (64, 32)]):
        w = (torch.randn(ci, co, 3, 3, generator=g) * 0.1).to(dtype).float().requires_grad_(True)
        x = torch.randn(N, ci, H // 2, W // 2, generator=g).to(dtype).float().requires_grad_(True)
        y = F.conv_transpose2d(x, w, None, stride=2, padding=1, output_padding=1)
        dy = (torch.randn(y.shape, generator=g) * 0.5).to(dtype).float()
        y.backward(dy)
        xe = cl(x.detach()).requires_grad_(True)
        we = w.detach().cuda().requires_grad_(True)
        ye = T.Deconv2dFn.apply(holder, f"u{tag}", dtype, xe, we)
        ye.backward(cl(dy))
        ulp = 2 ** -8 if dtype == torch.bfloat16 else 2 ** -11
        check_close(f"deconv {ci}->{co} y", cf(ye.detach()), y.detach(), rel_l2=2 * ulp)
        check_close(f"deconv {ci}->{co} dW", we.grad.cpu(), w.grad, rel_l2=3e-5)
        check_close(f"deconv {ci}->{co} dX", cf(xe.grad), x.grad, rel_l2=2 * ulp)
    # BatchNorm + ReLU (+ skip) with two groups = two separate module calls
    for C, relu, with_skip in ((128, "post", True), (32, "pre", False), (64, None, False)):
        bn_e, bn_r = torch.nn.BatchNorm2d(C).cuda().train(), torch.nn.BatchNorm2d(C).train()
        with torch.no_grad():
            bn_r.weight.copy_(torch.rand(C, generator=g) + 0.5); bn_r.bias.copy_(torch.randn(C, generator=g) * 0.2)
            bn_e.weight.copy_(bn_r.weight); bn_e.bias.copy_(bn_r.bias)
        y = torch.randn(4, C, 8, 12, generator=g).to(dtype).float().requires_grad_(True)
        sk = torch.randn(4, C, 8, 12, generator=g).to(dtype).float().requires_grad_(True) if with_skip else None
        outs = []
        for gi in range(2):        # the reference: one module call per view
            z = bn_r(y[2 * gi:2 * gi + 2])
            z = F.relu(z) if relu == "pre" else z
            z = z + sk[2 * gi:2 * gi + 2] if with_skip else z
            outs.append(F.relu(z) if relu == "post" else z)
        out = torch.cat(outs, 0)
        dout = torch.randn(out.shape, generator=g).to(dtype).float()
        out.backward(dout)
        ye = cl(y.detach()).requires_grad_(True)
        ske = cl(sk.detach()).requires_grad_(True) if with_skip else None
        oe = T.BnAct2dFn.apply(bn_e, 2, relu, ye, ske, bn_e.weight, bn_e.bias)
        oe.backward(cl(dout))
        ulp = 2 ** -8 if dtype == torch.bfloat16 else 2 ** -11
        check_close(f"bn{C} {relu} out", cf(oe.detach()), out.detach(), rel_l2=2 * ulp)
        check_close(f"bn{C} {relu} dy", cf(ye.grad), y.grad, rel_l2=4 * ulp)
        check_close(f"bn{C} {relu} dgamma", bn_e.weight.grad.cpu(), bn_r.weight.grad, rel_l2=4 * ulp)
        check_close(f"bn{C} {relu} dbeta", bn_e.bias.grad.cpu(), bn_r.bias.grad, rel_l2=4 * ulp)
        if with_skip:
            check_close(f"bn{C} {relu} dskip", cf(ske.grad), sk.grad, rel_l2=2 * ulp)
        check_close(f"bn{C} running_mean", bn_e.running_mean.cpu(), bn_r.running_mean, rel_l2=2 * ulp)
        check_close(f"bn{C} running_var", bn_e.running_var.cpu(), bn_r.running_var, rel_l2=2 * ulp)


@pytest.mark.parametrize("dtype", DT)
def test_vis_featext_forward_train_against_module_autograd(dtype):
    """FeatExt.forward_train (Vis-MVSNet's 2-D residual U-Net in train() mode built from the engine's layer nodes, all views in one
    pass with per-view BatchNorm statistics) against the module under PyTorch-ROCm autograd called once per view (fp32): the three
    feature maps, every parameter gradient (direction of the whole vector), the running statistics."""
    from wild_deep_mvs_amd import synthetic
    from wild_deep_mvs_amd.models.VisMVSNet.frontend import Frontend
    import copy
    net = Frontend()
    net.load_state_dict(synthetic.train_state_dict("vis", synthetic.template_of(net), seed=0))
    fa = net.model.feat_ext.cuda().train()
    fb = copy.deepcopy(fa)
    gen = torch.Generator().manual_seed(8)
    V, B, H, W = 3, 1, 64, 96
    imgs = [torch.rand(B, 3, H, W, generator=gen).cuda() for _ in range(V)]
    gouts = [[torch.randn(B, 32, H // s_, W // s_, generator=gen).cuda() for s_ in (8, 4, 2)] for _ in range(V)]
    outs = fa.forward_train(torch.cat(imgs, 0), V, dtype)
    torch.autograd.backward(outs, [torch.cat([gouts[v][k] for v in range(V)], 0).permute(0, 2, 3, 1).to(dtype).contiguous() for k in range(3)])
    refs = []
    for v in range(V):
        r = fb(imgs[v])
        torch.autograd.backward(r, gouts[v])
        refs.append([t_.detach() for t_ in r])
    torch.cuda.synchronize()
    bf = dtype == torch.bfloat16
    for k in range(3):
        check_close(f"FeatExt map {k}", outs[k].detach().float().permute(0, 3, 1, 2).cpu(), torch.cat([refs[v][k] for v in range(V)], 0).cpu(),
                    rel_l2=6e-2 if bf else 8e-3)
    worst, cos, rows = _grad_report(f"FeatExt.forward_train {dtype} vs module autograd", fa, {k: p.grad.detach().cpu() for k, p in fb.named_parameters()})
    assert cos >= (0.9 if bf else 0.99), (cos, rows)
    for (k, a_), (_, b_) in zip(fa.state_dict().items(), fb.state_dict().items()):
        if "running_" in k:
            check_close(f"stat {k}", a_.float().cpu(), b_.float().cpu(), rel_l2=6e-2 if bf else 8e-3)


@pytest.mark.parametrize("dtype", DT)
def test_vis_train_step_with_engine_extractor(dtype):
    """feature_engine_train = "pscv" on Vis-MVSNet: the whole training step (2-D extractor of all views included) on the engine against
    the step with the PyTorch-ROCm extractor on the same weights and scene: final depth, loss, direction of the full gradient."""
    from wild_deep_mvs_amd import synthetic
    from wild_deep_mvs_amd.models.VisMVSNet.frontend import Frontend
    import copy
    H, W, V, B = 64, 96, 3, 1
    kw = dict(depth_nums=[16, 8, 4], interval_scales=[2.0, 1.0, 0.5])
    na = Frontend()
    na.load_state_dict(synthetic.train_state_dict("vis", synthetic.template_of(na), seed=0))
    na.depth_nums, na.interval_scales = kw["depth_nums"], kw["interval_scales"]
    na = na.cuda().train()
    na.train_storage_dtype = dtype
    nb = copy.deepcopy(na)
    na.feature_engine_train = "pscv"
    scene = synthetic.make_scene(B, V, H, W, seed=4)
    args = [scene[k].cuda() for k in ("imgs", "K", "R", "t", "depth_min", "depth_max")]
    gt, mask = synthetic.train_target(scene, H // 2, W // 2)
    res = []
    for net in (na, nb):
        out = net(*args, **kw)
        loss = synthetic.vis_supervised_loss(out, gt.cuda(), mask.cuda(), args[4], args[5], V)
        loss.backward()
        res.append((out["depth"].detach().cpu(), float(loss.detach())))
    torch.cuda.synchronize()
    bf = dtype == torch.bfloat16
    check_close("final depth, engine extractor vs torch extractor", res[0][0], res[1][0], rel_l1=5e-2 if bf else 1e-2)
    assert abs(res[0][1] - res[1][1]) <= (1.5e-1 if bf else 4e-2) * abs(res[1][1]), (res[0][1], res[1][1])
    grads = {k: p.grad.detach().cpu() for k, p in nb.named_parameters() if p.grad is not None}
    worst, cos, rows = _grad_report(f"vis + engine FeatExt {dtype} vs torch extractor", na, grads)
    assert cos >= (0.6 if bf else 0.9), (cos, rows)


@pytest.mark.parametrize("dtype", DT)
def test_feature_pyramid_fn_against_module_autograd(dtype):
    """training.FeaturePyramidFn (CVP-MVSNet's nine-layer conv + LeakyReLU(0.1) tower on two pyramid levels, forward and backward on
    the engine, all views as one batch) against the same module under PyTorch-ROCm autograd (fp32): both levels' features and every
    weight / bias gradient (gradients of the two levels add up in the shared tower).  Nine stored 16-bit layers each way."""
    from wild_deep_mvs_amd import synthetic, training as T
    from wild_deep_mvs_amd.models.CVP_MVSNet.frontend import Frontend
    import copy
    net = Frontend()
    net.load_state_dict(synthetic.train_state_dict("cvp", synthetic.template_of(net), seed=0))
    fa = net.model.featurePyramid.cuda().train()
    fb = copy.deepcopy(fa)
    gen = torch.Generator().manual_seed(6)
    img = torch.rand(3, 3, 48, 80, generator=gen).cuda()
    gouts = [torch.randn(3, 16, 48, 80, generator=gen).cuda(), torch.randn(3, 16, 24, 40, generator=gen).cuda()]
    outs = T.FeaturePyramidFn.apply(fa, dtype, 2, img, *T.FeaturePyramidFn.params(fa))
    torch.autograd.backward(outs, [g.permute(0, 2, 3, 1).to(dtype).contiguous() for g in gouts])
    refs = fb(img, 2)
    torch.autograd.backward(refs, gouts)
    torch.cuda.synchronize()
    bf = dtype == torch.bfloat16
    for lv, (o, r) in enumerate(zip(outs, refs)):
        check_close(f"pyramid level {lv}", o.detach().float().permute(0, 3, 1, 2).cpu(), r.detach().cpu(), rel_l2=3e-2 if bf else 4e-3)
    o_grads = {k: p.grad.detach().cpu() for k, p in fb.named_parameters()}
    worst, cos, rows = _grad_report(f"FeaturePyramidFn {dtype} vs module autograd", fa, o_grads)
    assert cos >= (0.97 if bf else 0.995), (cos, rows)
    # only the finest level feeds a loss: the coarse level's gradient is None
    fa.zero_grad(); fb.zero_grad()
    outs = T.FeaturePyramidFn.apply(fa, dtype, 2, img, *T.FeaturePyramidFn.params(fa))
    outs[0].backward(gouts[0].permute(0, 2, 3, 1).to(dtype).contiguous())
    fb(img, 2)[0].backward(gouts[0])
    worst, cos, rows = _grad_report(f"FeaturePyramidFn {dtype}, finest level only", fa, {k: p.grad.detach().cpu() for k, p in fb.named_parameters()})
    assert cos >= (0.97 if bf else 0.995), (cos, rows)


@pytest.mark.parametrize("dtype", DT)
def test_cvp_train_step_with_engine_pyramid(dtype):
    """feature_engine_train = "pscv" on CVP-MVSNet: the whole training step (pyramid tower of all views included) on the engine,
    against the step with the PyTorch-ROCm tower on the same weights and scene: depth of both levels, loss, and the direction of
    the full gradient (a mixed-precision mode: nine more stored 16-bit layers in front of the sweep)."""
    from wild_deep_mvs_amd import synthetic
    from wild_deep_mvs_amd.models.CVP_MVSNet.frontend import Frontend
    import copy
    H, W, V, B = 64, 96, 3, 1
    na = Frontend()
    na.load_state_dict(synthetic.train_state_dict("cvp", synthetic.template_of(na), seed=0))
    na = na.cuda().train()
    na.train_storage_dtype = dtype
    nb = copy.deepcopy(na)
    na.feature_engine_train = "pscv"
    scene = synthetic.make_scene(B, V, H, W, seed=3)
    scene["t"] = scene["t"] * 8
    args = [scene[k].cuda() for k in ("imgs", "K", "R", "t", "depth_min", "depth_max")]
    gt, mask = synthetic.train_target(scene, H, W)
    res = []
    for net in (na, nb):
        out = net(*args, nscale=2)
        loss = synthetic.supervised_loss(out["depth"], gt.cuda(), mask.cuda(), scene["depth_min"].cuda(), scene["depth_max"].cuda())
        loss.backward()
        res.append((out, float(loss.detach())))
    torch.cuda.synchronize()
    bf = dtype == torch.bfloat16
    for lv in range(2):
        check_close(f"depth level {lv}, engine tower vs torch tower", res[0][0]["depth_est_list"][lv].detach().cpu(),
                    res[1][0]["depth_est_list"][lv].detach().cpu(), rel_l1=4e-2 if bf else 8e-3)
    assert abs(res[0][1] - res[1][1]) <= (1e-1 if bf else 3e-2) * abs(res[1][1]), (res[0][1], res[1][1])
    worst, cos, rows = _grad_report(f"cvp + FeaturePyramidFn {dtype} vs torch tower", na, {k: p.grad.detach().cpu() for k, p in nb.named_parameters() if p.grad is not None})
    assert cos >= (0.7 if bf else 0.95), (cos, rows)


@pytest.mark.parametrize("dtype", DT)
def test_mvsnet_train_step_with_engine_extractor(dtype):
    """feature_engine_train = "pscv": the whole MVSNet training step (2-D extractor included) on the engine.  A mixed-precision
    mode -- eight more stored 16-bit layers in front of the sweep -- so the bars against the fp32 oracle are those of the
    storage format, not of the path: depth, loss, direction of the full gradient."""
    from test_oracle_train import oracle_train_step
    from wild_deep_mvs_amd import synthetic
    from wild_deep_mvs_amd.models.MVSNet.model import MVSNet
    g = load_golden("mvsnet_train.npz")
    H, W, V, D, seed, scene_seed, B = [int(x) for x in g["meta"]]
    net = MVSNet("variance")
    net.load_state_dict(synthetic.train_state_dict("mvsnet", synthetic.template_of(net), seed=seed))
    net = net.cuda().train()
    net.num_depth, net.train_storage_dtype, net.feature_engine_train = D, dtype, "pscv"
    scene = synthetic.make_scene(B, V, H, W, seed=scene_seed)
    out = net(*[scene[k].cuda() for k in ("imgs", "K", "R", "t", "depth_min", "depth_max")])
    gt, mask = synthetic.train_target(scene, H // 4, W // 4)
    loss = synthetic.supervised_loss(out["depth"], gt.cuda(), mask.cuda(), scene["depth_min"].cuda(), scene["depth_max"].cuda())
    loss.backward()
    torch.cuda.synchronize()
    o_depth, o_loss, o_grads, _ = oracle_train_step("variance", H, W, V, D, seed, scene_seed, B)
    bf = dtype == torch.bfloat16
    check_close("depth vs oracle", out["depth"].detach().cpu(), o_depth, rel_l1=4e-2 if bf else 6e-3)
    assert abs(float(loss) - o_loss) <= (1e-1 if bf else 2e-2) * abs(o_loss), (float(loss), o_loss)
    worst, cos, rows = _grad_report(f"mvsnet + FeatureNetFn {dtype} vs fp32 oracle", net, o_grads)
    # (bf16: eight more 8-bit-significand layers in front of the chaotic tiny fixture; 0.80-0.81 depending on one-ulp details such as
    #  1 / sqrt against rsqrt in the BatchNorm bookkeeping -- a noise-level yardstick, the sharp checks are the per-kernel tests above)
    assert cos >= (0.7 if bf else 0.97), (cos, rows)


@pytest.mark.parametrize("case", ["headline", "zoomed_source"])
def test_warp_backward_tiled_vs_direct_kernel(case):
    """The LDS-privatised warp backward against the one-global-atomic-per-tap kernel (pscv_set_tuning("warp_bwd_direct")), two
    independent implementations, at sizes the CPU oracle does not reach: the headline sweep (5 views, 128x160x32 features,
    D = 192) and a source map at 4x the reference resolution, where a workgroup's texel bounding box exceeds the LDS patch
    and the clipped part takes the direct-atomic fallback."""
    from wild_deep_mvs_amd import _lib as L, ops, synthetic
    from oracle import mvsnet as O
    if case == "headline":
        V, h, w, hs, ws, D = 5, 128, 160, 128, 160, 192
    else:
        V, h, w, hs, ws, D = 3, 32, 40, 128, 160, 16
    scene = synthetic.make_scene(1, V, 4 * hs, 4 * ws, seed=3)
    K = scene["K"].clone()
    if case == "zoomed_source":
        K[:, 0, :2] /= 4                      # the reference camera sees the scene at a quarter of the sources' resolution
    proj, dv = O.mvsnet_cameras(K, scene["R"], scene["t"], scene["depth_min"], scene["depth_max"], D)
    gen = torch.Generator().manual_seed(12)
    dt = torch.float16
    ref = (torch.randn(1, h, w, 32, generator=gen) * 0.5).to(dt).cuda()
    srcs = [(torch.randn(1, hs, ws, 32, generator=gen) * 0.5).to(dt).cuda() for _ in range(V - 1)]
    cams = ops.proj_cams_device(proj.cuda().contiguous(), 0)
    depth = dv[:, 0].contiguous().cuda()
    g = (torch.randn(1, D, h, w, 32, generator=gen) * 0.1).to(dt).cuda()
    res = {}
    for direct in (0, 1):
        L.set_tuning("warp_bwd_direct", direct)
        try:
            res[direct] = ops.warp_cost_bwd(ref, srcs, cams, depth, g, geom=L.GEOM_PROJ, cost=L.COST_VARIANCE)
            torch.cuda.synchronize()
        finally:
            L.set_tuning("warp_bwd_direct", 0)
    check_close(f"{case}: d ref", res[0][0].cpu(), res[1][0].cpu(), rel_l2=1e-5)
    for v in range(V - 1):
        assert float(res[1][1][v].abs().max()) > 0
        # fixed-point LDS sums keep 21 bits below the workgroup's bound; both kernels then add fp32 atomics in arbitrary order
        check_close(f"{case}: d src{v}", res[0][1][v].cpu(), res[1][1][v].cpu(), rel_l2=1e-4)   # measured 2e-5
