"""GPU parity of the decoder / fine-tune / evaluation path: full UNet forward + one fine-tune step
against the golden vectors written from the reference UNet (tests/golden/g5_decoder.npz, tools/gen_golden.py
gen_decoder), and the head kernels (1x1 conv, softmax, KL_div, one-hot, arg-max, Dice counts) against the oracle."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import spcl_oracle as O
from tests._stability import oracle_sensitivity


def _unet(dtype=torch.float32):
    import spcl_amd  # noqa
    from spcl_amd.semi_seg.arch import UNet
    m = UNet(input_dim=1, num_classes=4, max_channel=128)
    m.load_state_dict(O.init_unet_state(1, 4, 128, seed=11), strict=True)
    m.cuda().train()
    m.set_compute_dtype(dtype)
    return m


def _relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(1e-12, np.abs(b).max())


def test_decoder_outputs_match_reference_fp32(golden):
    g = golden("g5_decoder.npz")
    x = torch.tensor(g["x"]).cuda()
    for until in ("Up_conv5", "Up_conv4", "Up_conv3", "Up_conv2"):
        y = _unet()(x, until=until)
        assert tuple(y.shape) == g[f"out/{until}"].shape
        assert _relerr(y.detach().float().cpu().numpy(), g[f"out/{until}"]) < 1e-3, until


def test_finetune_step_matches_reference_fp32(golden):
    """logits, supervised loss (softmax + KL_div on one-hot labels), every parameter gradient and BN buffer of one
    fine-tune step (new_epocher.py:268-277) vs the reference's modules; then eval-mode logits and arg-max."""
    from spcl_amd import functional as F
    g = golden("g5_decoder.npz")
    x, labels = torch.tensor(g["x"]).cuda(), torch.tensor(g["labels"]).cuda()
    m = _unet()
    logits = m(x)
    assert tuple(logits.shape) == (4, 4, 32, 32) and logits.dtype == torch.float32
    assert _relerr(logits.detach().cpu().numpy(), g["out/logits"]) < 1e-3
    loss = F.kl_div(F.softmax_classes(logits), F.one_hot_classes(labels, 4))
    assert abs(float(loss.detach()) - float(g["loss"])) < 2e-4 * abs(float(g["loss"]))
    loss.backward()
    params, bufs = dict(m.named_parameters()), dict(m.named_buffers())
    for k in g.files:
        if k.startswith("grad/"):
            assert params[k[5:]].grad is not None, k
            assert _relerr(params[k[5:]].grad.cpu().numpy(), g[k]) < 3e-3, (k, _relerr(params[k[5:]].grad.cpu().numpy(), g[k]))
        elif k.startswith("buf/"):
            np.testing.assert_allclose(bufs[k[4:]].cpu().numpy(), g[k], rtol=1e-4, atol=1e-6, err_msg=k)
    m.eval()
    with torch.no_grad():
        ev = m(x)
    assert _relerr(ev.cpu().numpy(), g["eval/logits"]) < 1e-3
    pred = F.argmax_classes(ev)
    assert (pred.cpu().numpy() == g["eval/pred"]).mean() > 0.999


def test_finetune_step_bf16_tracks_the_emulating_oracle(golden):
    """bf16 storage: loss within 2 % of the fp32 reference (SURVEY 8c bf16 tolerance)."""
    from spcl_amd import functional as F
    g = golden("g5_decoder.npz")
    x, labels = torch.tensor(g["x"]).cuda(), torch.tensor(g["labels"]).cuda()
    m = _unet(torch.bfloat16)
    loss = F.kl_div(F.softmax_classes(m(x)), F.one_hot_classes(labels, 4))
    assert abs(float(loss.detach()) - float(g["loss"])) < 2e-2 * abs(float(g["loss"]))
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("N,C,K,H,W", [(2, 16, 4, 9, 7), (1, 24, 3, 5, 5), (2, 40, 11, 4, 6)])
def test_conv1x1_forward_backward(dt, N, C, K, H, W):
    from spcl_amd import functional as F
    dtype = torch.float32 if dt == "f32" else torch.bfloat16
    g = torch.Generator().manual_seed(C * 7 + K)
    x = torch.randn(N, C, H, W, generator=g).to(dtype).float()
    w = torch.randn(K, C, 1, 1, generator=g) / C ** 0.5
    b = torch.randn(K, generator=g)
    r = torch.randn(N, K, H, W, generator=g)
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = torch.nn.functional.conv2d(xr, wr, br)
    (ref * r.double()).sum().backward()
    xd, wd, bd = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    out = F.conv1x1(xd, wd, bd, dtype)
    (out * r.cuda()).sum().backward()
    tol = 1e-5 if dt == "f32" else 6e-3
    assert _relerr(out.detach().cpu().numpy(), ref.detach().numpy()) < tol
    assert _relerr(xd.grad.cpu().numpy(), xr.grad.numpy()) < tol
    assert _relerr(wd.grad.cpu().numpy(), wr.grad.numpy()) < 1e-4
    assert _relerr(bd.grad.cpu().numpy(), br.grad.numpy()) < 1e-4


@pytest.mark.parametrize("K", [2, 4, 7])
def test_softmax_kl_onehot_argmax_dice_vs_oracle(K):
    from spcl_amd import functional as F
    g = torch.Generator().manual_seed(K)
    logits = torch.randn(3, K, 13, 10, generator=g) * 3
    labels = torch.randint(0, K, (3, 13, 10), generator=g)
    lr = logits.clone().requires_grad_(True)
    ref = O.finetune_loss(lr, labels)
    ref.backward()
    ld = logits.cuda().requires_grad_(True)
    prob = F.softmax_classes(ld)
    np.testing.assert_allclose(prob.detach().cpu().numpy(), logits.softmax(1).numpy(), rtol=1e-5, atol=1e-7)
    onehot = F.one_hot_classes(labels.cuda(), K)
    assert torch.equal(onehot.cpu().long(), O.class2one_hot(labels, K))
    loss = F.kl_div(prob, onehot)
    assert abs(float(loss.detach()) - float(ref.detach())) < 1e-5 * max(1.0, abs(float(ref.detach())))
    (loss * 1.7).backward()
    np.testing.assert_allclose(ld.grad.cpu().numpy(), 1.7 * lr.grad.numpy(), rtol=1e-4, atol=1e-8)
    pred = F.argmax_classes(ld.detach())
    assert torch.equal(pred.cpu(), logits.max(1)[1])
    inter, union = F.dice_counts(pred, labels.cuda(), K)
    ri, ru = O.dice_counts(logits.max(1)[1], labels, K)
    assert torch.equal(inter.cpu(), ri) and torch.equal(union.cpu(), ru)


@pytest.mark.parametrize("K,shape", [(4, (3, 13, 10)), (2, (2, 31, 17)), (7, (1, 64, 64)), (4, (5, 224, 224))])
def test_fused_sup_loss_is_the_separate_kernels_in_one_launch(K, shape):
    """`spcl_sup_loss_forward` (new_epocher.py:268-282 in one pass) against the oracle and against the chain it replaces
    (softmax -> one-hot -> KL_div -> backward, arg-max -> Dice counts): loss within float summation order, the gradient for a
    unit upstream gradient BIT-identical per pixel, the Dice counts equal; a non-unit upstream gradient scales it."""
    from spcl_amd import functional as F
    g = torch.Generator().manual_seed(K * 31 + shape[1])
    logits = torch.randn(shape[0], K, shape[1], shape[2], generator=g) * 3
    labels = torch.randint(0, K, shape, generator=g)
    lr = logits.clone().requires_grad_(True)
    ref = O.finetune_loss(lr, labels)
    ref.backward()
    # the chain of separate launches
    la = logits.cuda().requires_grad_(True)
    loss_a = F.kl_div(F.softmax_classes(la), F.one_hot_classes(labels.cuda(), K))
    loss_a.backward()
    ia, ua = F.dice_counts(F.argmax_classes(la.detach()), labels.cuda(), K)
    # fused
    lb = logits.cuda().requires_grad_(True)
    loss_b, (ib, ub) = F.sup_loss_kl_onehot(lb, labels.cuda())
    loss_b.backward()
    assert abs(float(loss_b) - float(ref.detach())) < 1e-5 * max(1.0, abs(float(ref.detach())))
    assert abs(float(loss_b) - float(loss_a)) < 2e-6 * max(1.0, abs(float(loss_a)))
    assert torch.equal(lb.grad, la.grad)  # the same per-pixel arithmetic in the same order
    np.testing.assert_allclose(lb.grad.cpu().numpy(), lr.grad.numpy(), rtol=1e-4, atol=1e-8)
    assert torch.equal(ib, ia) and torch.equal(ub, ua)
    ri, ru = O.dice_counts(logits.max(1)[1], labels, K)
    assert torch.equal(ib.cpu(), ri) and torch.equal(ub.cpu(), ru)
    lc = logits.cuda().requires_grad_(True)
    loss_c, _ = F.sup_loss_kl_onehot(lc, labels.cuda())
    (loss_c * 1.7).backward()
    np.testing.assert_allclose(lc.grad.cpu().numpy(), 1.7 * lr.grad.numpy(), rtol=1e-4, atol=1e-8)
    # the epocher's ``backward(gradient=ones)`` with the ones registered: no scaling pass, the same bits
    unit = F.register_unit_gradient(torch.ones((), device="cuda"))
    ld = logits.cuda().requires_grad_(True)
    loss_d, _ = F.sup_loss_kl_onehot(ld, labels.cuda())
    assert F.is_unit_gradient(unit) and not F.is_unit_gradient(torch.ones((), device="cuda"))
    loss_d.backward(gradient=unit)
    assert torch.equal(ld.grad, la.grad)
    two = torch.full((), 2.0, device="cuda")  # an unregistered upstream gradient still scales
    le = logits.cuda().requires_grad_(True)
    loss_e, _ = F.sup_loss_kl_onehot(le, labels.cuda())
    loss_e.backward(gradient=two)
    assert torch.equal(le.grad, la.grad * 2.0)
    # a caller that scales the registered tensor IN PLACE (loss scaling) breaks the promise: recognised by its version
    # counter, the backward multiplies by what the tensor holds now
    unit.mul_(3.0)
    assert not F.is_unit_gradient(unit)
    lf = logits.cuda().requires_grad_(True)
    loss_f, _ = F.sup_loss_kl_onehot(lf, labels.cuda())
    loss_f.backward(gradient=unit)
    assert torch.equal(lf.grad, la.grad * 3.0)
    ptr = unit.data_ptr()
    del unit  # held weakly: the registry's entry for THIS tensor dies with it (other tests' epochers may keep theirs alive)
    ent = F._UNIT_GRADIENTS.get(ptr)
    assert ent is None or ent[0]() is None or not F.is_unit_gradient(ent[0]())


def test_universal_dice_meter_matches_oracle():
    from spcl_amd.contrastyou.meters import UniversalDice
    g = torch.Generator().manual_seed(3)
    m = UniversalDice(4, report_axises=[1, 2, 3])
    inters, unions, names = [], [], []
    for it in range(3):
        pred = torch.randint(0, 4, (5, 12, 9), generator=g)
        tgt = torch.randint(0, 4, (5, 12, 9), generator=g)
        group = [f"patient{(it * 5 + i) // 4:03d}" for i in range(5)]
        m.add(pred.cuda(), tgt.cuda(), group_name=group)
        i, u = O.dice_counts(pred, tgt, 4)
        inters.append(i); unions.append(u); names += group
    mean, std = O.universal_dice(torch.cat(inters), torch.cat(unions), names)
    got_mean, got_std = m.value()
    np.testing.assert_allclose(got_mean.numpy(), mean.float().numpy(), rtol=1e-6)
    np.testing.assert_allclose(got_std.numpy(), std.float().numpy(), rtol=1e-5)
    s = m.summary()
    assert set(s) == {"DSC1", "DSC2", "DSC3", "DSC_mean"}
    assert abs(s["DSC_mean"] - float(mean[1:].mean())) < 1e-6


def test_finetune_trainer_runs_and_learns(tmp_path):
    """FineTuneTrainer mirror: two epochs on synthetic labelled data -- the supervised loss goes down, Dice / loss meters
    are populated, checkpoints are written, evaluation runs in eval mode without touching the BN running stats."""
    import spcl_amd  # noqa
    from spcl_amd.semi_seg.arch import UNet
    from spcl_amd.semi_seg.trainers import FineTuneTrainer
    from spcl_amd.contrastyou.losses.kl import KL_div
    from spcl_amd.synthetic import SyntheticLabeledLoader
    torch.manual_seed(0)
    model = UNet(input_dim=1, num_classes=4, max_channel=128)
    tra = SyntheticLabeledLoader(bs=4, size=32, device="cuda", seed=1)
    val = SyntheticLabeledLoader(bs=4, size=32, device="cuda", seed=1, twice=False, length=2)
    tr = FineTuneTrainer(model=model, labeled_loader=tra, val_loader=val, test_loader=val, criterion=KL_div(),
                         save_dir=str(tmp_path), max_epoch=2, num_batches=12, device="cuda", lr=2e-3, warmup_max=1,
                         multiplier=1)
    with pytest.raises(RuntimeError):
        tr.start_training()
    tr.init()
    hist = tr.start_training()
    assert len(hist) == 2
    l0, l1 = hist[0]["tra"]["semi"]["sup_loss"]["mean"], hist[1]["tra"]["semi"]["sup_loss"]["mean"]
    assert math.isfinite(l0) and l1 < l0
    assert 0.0 <= hist[1]["score"] <= 1.0 and "DSC_mean" in hist[1]["val"]["eval"]["dice"]
    assert (tmp_path / "last.pth").exists() and (tmp_path / "best.pth").exists()
    rm = model._Conv1.conv[1].running_mean.clone()
    tr.run_eval_epoch(model=model, loader=val)
    assert torch.equal(rm, model._Conv1.conv[1].running_mean)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("N,C,H,W", [(2, 16, 5, 7), (1, 48, 3, 4), (2, 8, 4, 4)])
def test_upsample2x_and_concat_match_torch(dt, N, C, H, W):
    from spcl_amd import functional as F
    dtype = torch.float32 if dt == "f32" else torch.bfloat16
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(N, C, H, W, generator=g).to(dtype)
    r = torch.randn(N, C, 2 * H, 2 * W, generator=g).to(dtype)
    xr = x.double().requires_grad_(True)
    ref = torch.nn.functional.interpolate(xr, scale_factor=2, mode="nearest")
    (ref * r.double()).sum().backward()
    xd = x.cuda().requires_grad_(True)
    y = F.upsample2x(xd, dtype)
    assert torch.equal(y.detach().cpu().double(), ref.detach())
    (y.float() * r.cuda().float()).sum().backward()
    assert _relerr(xd.grad.float().cpu().numpy(), xr.grad.numpy()) < (1e-6 if dt == "f32" else 6e-3)
    if C % 16 == 0:
        b = torch.randn(N, 2 * C, H, W, generator=g).to(dtype)
        r2 = torch.randn(N, 3 * C, H, W, generator=g).to(dtype)
        a_d, b_d = x.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
        out = F.concat_channels(a_d, b_d, dtype)
        assert torch.equal(out.detach().cpu(), torch.cat((x, b), 1))
        (out.float() * r2.cuda().float()).sum().backward()
        assert torch.equal(a_d.grad.cpu(), r2[:, :C]) and torch.equal(b_d.grad.cpu(), r2[:, C:])


SPLIT_SLACK_CAP = 5e-2


@pytest.mark.parametrize("f32_products", ["exact", "split"])
def test_full_unet_base_width_vs_oracle_fp32(f32_products):
    """max_channel=256 (all channel counts multiples of 16: HIP concatenation / upsample path) on a small image: logits
    and a sample of gradients against the CPU oracle's full UNet.  ``exact``: the f32 convolutions on the exact-f32 MFMA;
    ``split`` (the default): gradients to the oracle's own sensitivity to fp32 rounding noise where that is larger than the
    bar (tests/_stability.py)."""
    import spcl_amd  # noqa
    from spcl_amd import native as _nat
    _nat.call("spcl_conv_set_f32_split", 1 if f32_products == "split" else 0)
    try:
        _full_unet_base_width_body(f32_products)
    finally:
        _nat.call("spcl_conv_set_f32_split", 1)


def _full_unet_base_width_body(f32_products):
    import spcl_amd  # noqa
    from spcl_amd import functional as F
    from spcl_amd.semi_seg.arch import UNet
    sd = O.init_unet_state(1, 4, 256, seed=5)
    m = UNet(input_dim=1, num_classes=4, max_channel=256)
    m.load_state_dict(sd, strict=True)
    m.cuda().train()
    m.set_compute_dtype(torch.float32)
    g = torch.Generator().manual_seed(6)
    x = torch.rand(2, 1, 64, 64, generator=g)  # 4x4 at Conv5: batch statistics over 32 values per channel
    labels = torch.randint(0, 4, (2, 64, 64), generator=g)
    keys = ("_Deconv_1x1.weight", "_Up_conv2.conv.0.weight", "_Up2.up.1.weight", "_Up_conv5.conv.3.weight",
            "_Up5.up.2.weight", "_Conv5.conv.0.weight", "_Conv1.conv.0.weight", "_Conv3.conv.4.bias")

    def oracle(x_):
        sdo_ = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
                for k, v in sd.items()}
        ref_logits_ = O.unet_forward(x_, sdo_, None, train=True)
        ref_loss_ = O.finetune_loss(ref_logits_, labels)
        ref_loss_.backward()
        return ref_logits_, ref_loss_, sdo_

    ref_logits, ref_loss, sdo = oracle(x)
    slack = 0.0
    if f32_products == "split":
        # capped (a bar derived from the failure it excuses pins nothing): what the cap cannot cover is
        # covered without ties by test_f32_split_and_exact_products_agree_call_by_call and, against fp64 on a network whose
        # statistics span > 1 000 values, by test_full_unet_wide_statistics_vs_fp64_oracle_fp32 below
        slack = min(SPLIT_SLACK_CAP, 3.0 * oracle_sensitivity((x,), lambda x_: {k: oracle(x_)[2][k].grad.numpy() for k in keys}))
    logits = m(x.cuda())
    assert _relerr(logits.detach().cpu().numpy(), ref_logits.detach().numpy()) < 2e-3
    loss = F.kl_div(F.softmax_classes(logits), F.one_hot_classes(labels.cuda(), 4))
    assert abs(float(loss.detach()) - float(ref_loss.detach())) < 1e-3 * abs(float(ref_loss.detach()))
    loss.backward()
    params = dict(m.named_parameters())
    for k in keys:
        # tiny batch statistics in the deepest layers amplify fp32 summation-order differences: 3e-2 of max
        assert _relerr(params[k].grad.cpu().numpy(), sdo[k].grad.numpy()) < max(3e-2, slack), (k, _relerr(
            params[k].grad.cpu().numpy(), sdo[k].grad.numpy()), slack)


def test_f32_split_and_exact_products_agree_call_by_call():
    """Tie-independent half of the f32 product-mode parity: EVERY convolution, concatenating convolution and weight-gradient
    call of one fine-tune step of the small full UNet (forward and backward), run under both modes ON THE SAME INPUTS -- no
    ReLU / max-pool decision lies between the two results of a call, so they must agree to f32 rounding: outputs and
    BatchNorm statistics rows within 3e-6 of the largest element (six bf16 products per element drop terms below 2^-24)."""
    import spcl_amd  # noqa
    from spcl_amd import functional as Fn
    from spcl_amd import native as _nat
    from spcl_amd.semi_seg.arch import UNet
    sd = O.init_unet_state(1, 4, 256, seed=5)
    m = UNet(input_dim=1, num_classes=4, max_channel=256)
    m.load_state_dict(sd, strict=True)
    m.cuda().train()
    m.set_compute_dtype(torch.float32)
    g = torch.Generator().manual_seed(6)
    x = torch.rand(2, 1, 64, 64, generator=g)
    labels = torch.randint(0, 4, (2, 64, 64), generator=g)
    seen, worst = [], [("", 0.0)]

    def rel(a, b):
        return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))

    def both(fn, name):
        def wrapped(*a, **k):
            out = fn(*a, **k)
            _nat.call("spcl_conv_set_f32_split", 0)
            try:
                ref = fn(*a, **k)
            finally:
                _nat.call("spcl_conv_set_f32_split", 1)
            o, r = (out[0] if isinstance(out, tuple) else out), (ref[0] if isinstance(ref, tuple) else ref)
            if torch.is_tensor(o) and o.dtype == torch.float32:
                e = rel(o, r)
                seen.append((name, tuple(o.shape), e))
                worst[0] = max(worst[0], (f"{name}{tuple(o.shape)}", e), key=lambda t: t[1])
            if isinstance(out, tuple) and len(out) > 1 and torch.is_tensor(out[1]) and hasattr(out[1], "ntiles"):
                cs = out[0].shape[-1]
                sa = out[1][:out[1].ntiles * 3 * cs].view(-1, 3, cs)
                sb = ref[1][:ref[1].ntiles * 3 * cs].view(-1, 3, cs)
                for comp, nm in ((1, "mean"), (2, "M2")):
                    e = rel(sa[:, comp], sb[:, comp])
                    seen.append((name + "." + nm, tuple(sa.shape), e))
                    worst[0] = max(worst[0], (f"{name}.{nm}{tuple(o.shape)}", e), key=lambda t: t[1])
            return out
        return wrapped

    names = [n for n in ("_conv", "_wgrad", "_conv_cat", "_wgrad_up2", "_wgrad_cat") if hasattr(Fn, n)]
    saved = {n: getattr(Fn, n) for n in names}
    _nat.call("spcl_conv_set_f32_split", 1)
    try:
        for n in names:
            setattr(Fn, n, both(saved[n], n))
        logits = m(x.cuda())
        loss = Fn.kl_div(Fn.softmax_classes(logits), Fn.one_hot_classes(labels.cuda(), 4))
        loss.backward()
    finally:
        for n in names:
            setattr(Fn, n, saved[n])
        _nat.call("spcl_conv_set_f32_split", 1)
    kinds = {n for n, _, _ in seen}
    assert len(seen) >= 60 and {"_conv", "_wgrad"} <= kinds, (len(seen), kinds)  # 23 convolutions x (forward, dgrad, wgrad)
    assert worst[0][1] <= 3e-6, worst[0]  # (measured 1.96e-6 on the 8 x 8 x 128 output of Conv4)
    print(f"\n[f32 modes] {len(seen)} calls compared, worst {worst[0][0]}: {worst[0][1]:.2e}")


def test_full_unet_wide_statistics_vs_fp64_oracle_fp32():
    """The other tie-independent half: a fine-tune step of the full UNet whose SMALLEST batch statistic spans 16 x 8 x 8 =
    1 024 values (16 slices of 128 x 128), default f32 product mode (split-bf16), against the oracle evaluated in fp64, with
    the fp32 CPU oracle (the reference's arithmetic) beside it as the yardstick.  Logits 1e-4 and loss 1e-5 from fp64.
    Gradients: at a random initialisation on random label maps they are ill-conditioned whatever computes them -- the fp32 CPU
    oracle itself sits 2e-3 .. 1e-2 (relative L2) from fp64 -- so the bar is the one tests/test_gpu_configs.py holds the
    pre-train step to: per tensor, the device's relative L2 distance to fp64 <= 2.5 x the fp32 oracle's own + 1e-3 (two samples
    of the same rounding noise), with no largest-element term and no tie slack: with this many values per statistic no single
    ReLU / max-pool decision moves a tensor."""
    import spcl_amd  # noqa
    from spcl_amd import functional as F
    from spcl_amd import native as _nat
    from spcl_amd.semi_seg.arch import UNet
    assert _nat.call("spcl_conv_get_f32_split") == 1
    sd = O.init_unet_state(1, 4, 256, seed=15)
    m = UNet(input_dim=1, num_classes=4, max_channel=256)
    m.load_state_dict(sd, strict=True)
    m.cuda().train()
    m.set_compute_dtype(torch.float32)
    g = torch.Generator().manual_seed(16)
    x = torch.rand(16, 1, 128, 128, generator=g)
    labels = torch.randint(0, 4, (16, 128, 128), generator=g)
    keys = ("_Deconv_1x1.weight", "_Up_conv2.conv.0.weight", "_Up2.up.1.weight", "_Up_conv5.conv.3.weight",
            "_Up5.up.2.weight", "_Conv5.conv.0.weight", "_Conv1.conv.0.weight", "_Conv3.conv.4.bias", "_Conv4.conv.3.weight",
            "_Up_conv3.conv.1.weight")

    def oracle(dt):
        sdo_ = {k: (v.to(dt).requires_grad_(True) if v.is_floating_point() and "running" not in k else
                    (v.to(dt) if v.is_floating_point() else v.clone())) for k, v in sd.items()}
        logits_ = O.unet_forward(x.to(dt), sdo_, None, train=True)
        loss_ = O.finetune_loss(logits_, labels)
        loss_.backward()
        return logits_.detach(), float(loss_.detach()), {k: sdo_[k].grad.double().numpy() for k in keys}

    threads = torch.get_num_threads()
    torch.set_num_threads(min(16, threads))
    try:
        logits64, loss64, g64 = oracle(torch.float64)
        _, _, g32 = oracle(torch.float32)
    finally:
        torch.set_num_threads(threads)
    logits = m(x.cuda())
    assert _relerr(logits.detach().cpu().numpy(), logits64.numpy()) < 1e-4
    loss = F.kl_div(F.softmax_classes(logits), F.one_hot_classes(labels.cuda(), 4))
    assert abs(float(loss.detach()) - loss64) < 1e-5 * abs(loss64)
    loss.backward()
    params = dict(m.named_parameters())

    def l2(a, b):
        return float(np.linalg.norm(a - b) / max(1e-30, np.linalg.norm(b)))
    table = {k: (l2(params[k].grad.double().cpu().numpy(), g64[k]), l2(g32[k], g64[k])) for k in keys}
    for k, (dev, cpu) in table.items():
        assert dev < 2.5 * cpu + 1e-3, (k, dev, cpu)
    print("\n[wide statistics, split-bf16] relative L2 to fp64, device | fp32 CPU oracle: " +
          "  ".join(f"{k} {d:.1e}|{c:.1e}" for k, (d, c) in table.items()))


@pytest.mark.parametrize("N,S", [(2, 224), (3, 112), (2, 128), (1, 256)])
def test_in_place_decoder_data_movement_is_the_copying_path_bit_for_bit(N, S, monkeypatch):
    """The decoder's concatenations / upsamples folded into their producers and consumers -- halves of one buffer written in
    place (>= 32 channels), the 16 + 16 channel level read from its two tensors by the convolution and its weight gradient
    (spcl_conv3x3_forward_cat / spcl_conv3x3_wgrad_cat), activations written x2-upsampled -- against the same network with
    the copying kernels (concat2, upsample2x): logits, loss and EVERY parameter gradient bit for bit (bf16)."""
    import spcl_amd  # noqa
    from spcl_amd import functional as F
    from spcl_amd.semi_seg.arch import UNet, unet as unet_mod
    sd = O.init_unet_state(1, 4, 256, seed=11)
    g = torch.Generator().manual_seed(12)
    x = torch.rand(N, 1, S, S, generator=g).cuda()
    labels = torch.randint(0, 4, (N, S, S), generator=g).cuda()

    def run(in_place, pair_first=True):
        # (the BatchNorm sums as per-tile rows in both forms: the accumulator blocks -- functional._BN_ACC -- are offered to the
        # plain one-tensor convolution only, so the copying form would take them where the two-tensor form cannot, and sums
        # formed in another order are not the same bits.  tests/test_gpu_bn_acc.py compares the two statistics routes.)
        monkeypatch.setattr(F, "_BN_ACC", False)
        monkeypatch.setattr(unet_mod, "_VIRTUAL_CAT", in_place)
        monkeypatch.setattr(unet_mod, "_FUSED_UPSAMPLE", in_place)
        monkeypatch.setattr(F, "_CONV_CAT", in_place)
        monkeypatch.setattr(unet_mod, "_CAT_PAIR_FIRST", pair_first)
        monkeypatch.setattr(unet_mod, "_CAT_PAIR_MAXC", 1024)  # (the wide levels too: slab-wise / block-wise two-tensor reads)
        monkeypatch.setattr(unet_mod, "_LAZY_HEAD", False)  # (its BatchNorm-backward sums have their own order: next test)
        monkeypatch.setattr(unet_mod, "_SPLIT_BNSTATS", False)  # (sums in another order: test_split_dgrad_... below)
        monkeypatch.setattr(unet_mod, "_LAZY_UP", in_place and pair_first)  # (the up-convolutions' BN + ReLU in the consumers' loaders)
        m = UNet(input_dim=1, num_classes=4, max_channel=256)
        m.load_state_dict(sd, strict=True)
        m.cuda().train()
        m.set_compute_dtype(torch.bfloat16)
        copies, bufs = [], []
        real_buf = F.cat_buffer
        monkeypatch.setattr(F, "cat_buffer", lambda *a, **k: (bufs.append(1), real_buf(*a, **k))[1])
        real_cat = F.concat_channels
        monkeypatch.setattr(F, "concat_channels", lambda *a, **k: (copies.append(1), real_cat(*a, **k))[1])
        logits = m(x)
        monkeypatch.setattr(F, "concat_channels", real_cat)
        monkeypatch.setattr(F, "cat_buffer", real_buf)
        if S in (224, 112):  # (14-column-tileable sizes: every level has its in-place form)
            assert len(copies) == (0 if in_place else 4)  # no level falls back to the copying concatenation
            assert len(bufs) == (0 if (pair_first or not in_place) else 3)  # the buffer only where the two-tensor read is not asked for
        else:  # 128 / 256: shifted tiles at the top, the band-GEMM kernels below -- whatever mix results must agree
            assert len(copies) == 4 if not in_place else len(copies) <= 4
        loss, _ = F.sup_loss_kl_onehot(logits, labels)
        loss.backward()
        torch.cuda.synchronize()
        return logits.detach().clone(), loss.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}

    lb, lossb, gb = run(False)
    for pair_first in (True, False):  # every level read from its two tensors / the >= 32-channel levels as halves of one buffer
        la, lossa, ga = run(True, pair_first)
        assert torch.equal(la, lb) and torch.equal(lossa, lossb)
        assert set(ga) == set(gb)
        for k in ga:
            assert torch.equal(ga[k], gb[k]), (pair_first, k)
            assert float(ga[k].abs().max()) > 0, k
    # and the narrow level really took the two-tensor convolution
    monkeypatch.setattr(F, "_CONV_CAT", True)
    if S in (224, 112):
        a = torch.zeros(N, 16, S, S, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        assert F.cat_pair_supported(a, a.clone(memory_format=torch.channels_last), 16, torch.bfloat16)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_last_activation_folded_into_the_head_matches_the_written_one(dt, monkeypatch):
    """`spcl_conv1x1_forward_bn` / `_backward_bn`: the decoder's last BatchNorm + ReLU applied inside the 1x1 head (no
    activation tensor, the BatchNorm-backward sums left by the head's backward) against the same network writing the
    activation: logits and loss bit for bit (the activation is rounded exactly as the writer stores it), gradients to the
    summation order of the partial sums; eval mode too."""
    import spcl_amd  # noqa
    from spcl_amd import functional as F
    from spcl_amd.semi_seg.arch import UNet, unet as unet_mod
    sd = O.init_unet_state(1, 4, 256, seed=21)
    g = torch.Generator().manual_seed(22)
    x = torch.rand(3, 1, 112, 112, generator=g).cuda()
    labels = torch.randint(0, 4, (3, 112, 112), generator=g).cuda()

    def run(lazy, train=True):
        monkeypatch.setattr(unet_mod, "_LAZY_HEAD", lazy)
        m = UNet(input_dim=1, num_classes=4, max_channel=256)
        m.load_state_dict(sd, strict=True)
        m.cuda().train(train)
        m.set_compute_dtype(dt)
        used = []
        real = F.conv1x1_bn
        monkeypatch.setattr(F, "conv1x1_bn", lambda *a, **k: (used.append(1), real(*a, **k))[1])
        logits = m(x)
        monkeypatch.setattr(F, "conv1x1_bn", real)
        assert len(used) == (1 if lazy else 0)
        loss, _ = F.sup_loss_kl_onehot(logits, labels)
        loss.backward()
        torch.cuda.synchronize()
        return (logits.detach().clone(), loss.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()},
                {k: v.clone() for k, v in m.state_dict().items() if "running" in k})

    for train in (True, False):
        la, lossa, ga, ra = run(True, train)
        lb, lossb, gb, rb = run(False, train)
        assert torch.equal(la, lb) and torch.equal(lossa, lossb)
        for k in rb:
            assert torch.equal(ra[k], rb[k]), k
        # fp32: the sums' order only; bf16: that order's last-bit differences re-rounded by every layer below (the encoder's
        # first layers see ~1 % of their largest entry -- the noise floor test_gpu_configs.py documents for bf16 gradients)
        tol = 2e-5 if dt == torch.float32 else 3e-2
        for k in gb:
            ref = gb[k].float()
            err = float((ga[k].float() - ref).norm()) / max(float(ref.norm()), 1e-20)
            assert err < tol, (k, err)


def test_split_dgrad_leaves_the_up_convolutions_bn_sums(monkeypatch):
    """`spcl_conv3x3_dgrad_split_bnstats`: the two-tensor level's input-gradient kernel also leaves the BatchNorm-backward sums
    of the up-convolution (whose activation is its second input): same network with and without -- logits identical, every
    gradient to the summation order of those sums (fp32: 2e-5 of its norm)."""
    import spcl_amd  # noqa
    from spcl_amd import functional as F
    from spcl_amd.semi_seg.arch import UNet, unet as unet_mod
    sd = O.init_unet_state(1, 4, 256, seed=31)
    g = torch.Generator().manual_seed(32)
    x = torch.rand(2, 1, 224, 224, generator=g).cuda()
    labels = torch.randint(0, 4, (2, 224, 224), generator=g).cuda()

    def run(on):
        monkeypatch.setattr(unet_mod, "_SPLIT_BNSTATS", on)
        monkeypatch.setattr(unet_mod, "_LAZY_HEAD", False)
        m = UNet(input_dim=1, num_classes=4, max_channel=256)
        m.load_state_dict(sd, strict=True)
        m.cuda().train()
        m.set_compute_dtype(torch.bfloat16)
        used = []
        real = F._n.call

        def spy(name, *a):
            if name == "spcl_conv3x3_dgrad_split_bnstats":
                used.append(1)
            return real(name, *a)
        monkeypatch.setattr(F._n, "call", spy)
        logits = m(x)
        loss, _ = F.sup_loss_kl_onehot(logits, labels)
        loss.backward()
        torch.cuda.synchronize()
        monkeypatch.setattr(F._n, "call", real)
        assert len(used) == (2 if on else 0)  # the 16- and the 32-channel level
        return logits.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}

    la, ga = run(True)
    lb, gb = run(False)
    assert torch.equal(la, lb)
    for k in gb:
        ref = gb[k].float()
        err = float((ga[k].float() - ref).norm()) / max(float(ref.norm()), 1e-20)
        assert err < 3e-2, (k, err)  # bf16 storage: the sums' last bits re-rounded by every layer below (see the test above)
    # the layers ABOVE the first affected BatchNorm see identical gradients
    for k in ("_Deconv_1x1.weight", "_Up_conv2.conv.3.weight", "_Up_conv2.conv.0.weight"):
        assert torch.equal(ga[k], gb[k]), k


def test_up_link_takes_a_second_contribution():
    """UpLink: the up-convolution that reads a block's activation at half resolution leaves the 2 x 2 gradient sums to that
    block and sends a shared tensor of zeros through autograd in their place -- if anything else contributed to the
    activation's gradient the block receives THAT contribution (zeros + it) and adds the sums itself (it used to
    send unwritten memory and refuse); with the single consumer the result equals the upsampled-tensor path bit for bit."""
    import spcl_amd  # noqa
    from spcl_amd import functional as F
    from spcl_amd.semi_seg.arch.unet import _ConvBlock, _UpConv
    torch.manual_seed(3)
    blk = _ConvBlock(32, 64).cuda().train()
    up = _UpConv(64, 32).cuda().train()
    for m in (blk, up):
        m._compute_dtype = torch.bfloat16
    x = torch.rand(2, 32, 28, 28, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    assert F.up_in_shape_ok(2, 64, 56, 56, 32, torch.bfloat16)

    def run(extra, virtual):
        for m in (blk, up):
            m.zero_grad(set_to_none=True)
        link = F.UpLink() if virtual else None
        blk._plan, blk._up_link = (True, False), link
        a = blk(x)
        o = up(a, virtual_up=virtual, up_link=link)
        loss = o.float().square().mean() + (a.float().mean() if extra else 0.0)
        loss.backward()
        torch.cuda.synchronize()
        return [p.grad.clone() for p in list(blk.parameters()) + list(up.parameters())]

    ga, gb = run(False, True), run(False, False)
    for a, b in zip(ga, gb):
        assert torch.equal(a, b)
    gc, gd = run(True, True), run(True, False)  # a second consumer: the linked form recovers, equal to the ordinary path
    for a, b in zip(gc, gd):
        assert float((a.float() - b.float()).norm()) <= 1e-2 * float(b.float().norm()) + 1e-12  # (bf16 gradient sums, another order)
    for a, b in zip(ga, run(False, True)):  # ... and the shared zeros are still zeros: the single-consumer result is unchanged
        assert torch.equal(a, b)
