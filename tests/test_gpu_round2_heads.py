"""Round-2 additions on the GPU against the reference's golden vectors (tests/golden/g6_round2.npz) and the oracle:
SupConLoss1(exclude_other_pos=True) (contrast_loss3.py:97-100), ProjectionHead(pool_name="adaptive_max"),
DenseProjectionHead (projectors/heads.py:96-120), the adaptive pooling kernels, and the dense InfoNCE hook
(semi_seg/hooks/infonce.py:201-241)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import spcl_oracle as O
from tests._stability import oracle_sensitivity
from tests.test_oracle_golden import labels_of


def test_exclude_other_pos_golden_and_oracle(golden):
    import spcl_amd  # noqa
    from spcl_amd.contrastyou.losses.contrast_loss3 import SupConLoss1
    g = golden("g6_round2.npz")
    for key in g["xpos/cases"]:
        key = str(key)
        n, d, lname = int(key.split("/")[1].split("_")[0][1:]), int(key.split("_")[1][1:]), key.split("_", 2)[2]
        z1 = torch.tensor(g[f"xpos/n{n}_d{d}/z1"], device="cuda", requires_grad=True)
        z2 = torch.tensor(g[f"xpos/n{n}_d{d}/z2"], device="cuda", requires_grad=True)
        crit = SupConLoss1(temperature=0.07, exclude_other_pos=True)
        loss = crit(z1, z2, target=labels_of(lname, n))
        loss.backward()
        np.testing.assert_allclose(loss.item(), g[f"{key}/loss"], rtol=1e-4, err_msg=key)
        np.testing.assert_allclose(z1.grad.cpu().numpy(), g[f"{key}/dz1"], rtol=2e-3, atol=1e-5, err_msg=key)
        np.testing.assert_allclose(z2.grad.cpu().numpy(), g[f"{key}/dz2"], rtol=2e-3, atol=1e-5, err_msg=key)
        assert crit.pos_mask.shape == (2 * n, 2 * n)  # the taps are still there
    # larger, against the oracle; SimCLR (no target) and an explicit mask too
    gen = torch.Generator().manual_seed(5)
    z1 = torch.nn.functional.normalize(torch.randn(300, 128, generator=gen), dim=1)
    z2 = torch.nn.functional.normalize(torch.randn(300, 128, generator=gen), dim=1)
    mask = (torch.arange(300)[:, None] % 7 == torch.arange(300)[None, :] % 7).float()
    for kw in (dict(target=[i % 5 for i in range(300)]), dict(), dict(mask=mask)):
        a, b = z1.clone().requires_grad_(True), z2.clone().requires_grad_(True)
        ref = O.supcon_loss_exclude_other_pos(a, b, kw.get("target"), kw.get("mask"))
        ref.backward()
        x, y = z1.cuda().requires_grad_(True), z2.cuda().requires_grad_(True)
        kwg = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in kw.items()}
        loss = SupConLoss1(exclude_other_pos=True)(x, y, **kwg)
        loss.backward()
        np.testing.assert_allclose(loss.item(), ref.item(), rtol=1e-4)
        scale = float(a.grad.abs().max())
        np.testing.assert_allclose(x.grad.cpu().numpy(), a.grad.numpy(), rtol=2e-3, atol=2e-4 * scale)
    with pytest.raises(AssertionError):  # not unit-norm (contrast_loss3.py:62)
        SupConLoss1(exclude_other_pos=True)(2 * z1.cuda(), z2.cuda(), target=[i % 5 for i in range(300)])


def _load_head(head, g, tag):
    sd = {k[len(tag) + 7:]: torch.tensor(g[k]) for k in g.files if k.startswith(f"{tag}/param/")}
    head.load_state_dict(sd, strict=True)
    return head.cuda()


def test_adaptive_max_projection_head_golden(golden):
    import spcl_amd  # noqa
    from spcl_amd.contrastyou.projectors.heads import ProjectionHead
    g = golden("g6_round2.npz")
    head = _load_head(ProjectionHead(input_dim=32, hidden_dim=24, output_dim=16, head_type="mlp", normalize=True,
                                     pool_name="adaptive_max"), g, "maxhead")
    x = torch.tensor(g["maxhead/x"], device="cuda", requires_grad=True)
    z = head(x)
    (z * torch.tensor(g["maxhead/r"], device="cuda")).sum().backward()
    np.testing.assert_allclose(z.detach().cpu().numpy(), g["maxhead/z"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["maxhead/dx"], rtol=1e-3, atol=1e-6)
    for k, p in head.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), g[f"maxhead/grad/{k}"], rtol=1e-3, atol=1e-6, err_msg=k)
    # a spatial size other than (1, 1) constructs, and fails in forward as the reference's Flatten -> Linear does
    bad = ProjectionHead(input_dim=32, output_dim=16, head_type="mlp", normalize=True, spatial_size=(2, 2)).cuda()
    with pytest.raises(RuntimeError):
        bad(x)


@pytest.mark.parametrize("tag,kw", [("dense_mlp", dict(head_type="mlp", pool_name="adaptive_avg", spatial_size=(5, 4))),
                                    ("dense_lin", dict(head_type="linear", pool_name="adaptive_max", spatial_size=(3, 3)))])
def test_dense_projection_head_golden(golden, tag, kw):
    import spcl_amd  # noqa
    from spcl_amd.contrastyou.projectors.heads import DenseProjectionHead
    g = golden("g6_round2.npz")
    head = _load_head(DenseProjectionHead(input_dim=16, hidden_dim=24, output_dim=12, normalize=True, **kw), g, tag)
    assert sorted(head.state_dict()) == sorted(k[len(tag) + 7:] for k in g.files if k.startswith(f"{tag}/param/"))
    x = torch.tensor(g[f"{tag}/x"], device="cuda", requires_grad=True)
    z = head(x)
    assert tuple(z.shape) == g[f"{tag}/z"].shape
    (z * torch.tensor(g[f"{tag}/r"], device="cuda")).sum().backward()
    np.testing.assert_allclose(z.detach().cpu().numpy(), g[f"{tag}/z"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(x.grad.cpu().numpy(), g[f"{tag}/dx"], rtol=1e-3, atol=2e-6)
    for k, p in head.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), g[f"{tag}/grad/{k}"], rtol=1e-3, atol=2e-6, err_msg=k)


@pytest.mark.parametrize("mode", ["avg", "max"])
@pytest.mark.parametrize("shape,out", [((2, 16, 14, 11), (5, 4)), ((1, 48, 7, 7), (10, 10)), ((3, 32, 28, 28), (1, 1)),
                                       ((2, 64, 9, 13), (9, 13))])
def test_adaptive_pool_vs_oracle_bf16_and_f32(mode, shape, out):
    import spcl_amd  # noqa
    from spcl_amd import functional as F_
    gen = torch.Generator().manual_seed(sum(shape))
    for dt in (torch.float32, torch.bfloat16):
        x = torch.randn(*shape, generator=gen).to(dt).float()
        xr = x.clone().requires_grad_(True)
        ref = O.adaptive_pool2d(xr, out, mode)
        r = torch.randn(ref.shape, generator=gen)
        (ref * r).sum().backward()
        xg = x.to(dt).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        got = F_.adaptive_pool2d(xg, out, mode)
        (got * r.cuda()).sum().backward()
        np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-5, atol=1e-6)
        tol = 1e-6 if dt == torch.float32 else 8e-3
        np.testing.assert_allclose(xg.grad.float().cpu().numpy(), xr.grad.numpy(), rtol=tol, atol=tol)


@pytest.mark.parametrize("f32_products", ["exact", "split"])
def test_dense_infonce_hook_step_vs_oracle_fp32(f32_products):
    """INFONCEHook on a decoder feature (Up_conv3): encoder + decoder forward, dense head, 5 points per slice drawn under
    FixRandomSeed, SupConLoss1 with every point its own class -- loss and the decoder / head gradients against the oracle
    (the encoder is frozen, as main_pretrain_decoder.py:66-69 arranges).  ``exact``: the f32 convolutions on the exact-f32
    MFMA, gradients to 5e-3; ``split`` (the default mode, three bf16 pieces per operand): to the oracle's own sensitivity to
    fp32 rounding noise where that is larger (tests/_stability.py: ReLU / max-pool decisions at a tie)."""
    import spcl_amd  # noqa
    from spcl_amd import native as _nat
    _nat.call("spcl_conv_set_f32_split", 1 if f32_products == "split" else 0)
    try:
        _dense_infonce_hook_step_body(f32_products)
    finally:
        _nat.call("spcl_conv_set_f32_split", 1)


def _dense_infonce_hook_step_body(f32_products):
    import spcl_amd  # noqa
    from spcl_amd import ddp
    from spcl_amd.semi_seg.arch import UNet
    from spcl_amd.semi_seg.epochers import PretrainDecoderEpocher
    from spcl_amd.semi_seg.hooks import create_infonce_hooks, feature_until_from_hooks
    from spcl_amd.synthetic import acdc_like_meta
    mc, bs, size, seed = 128, 4, 32, 13
    net = UNet(input_dim=1, num_classes=4, max_channel=mc)
    sd = O.init_unet_state(1, 4, mc, seed=23)
    net.load_state_dict(sd, strict=True)
    net.cuda().train()
    hook = create_infonce_hooks(model=net, feature_names="Up_conv3", weights=0.5, contrast_ons="partition",
                                data_name="acdc").cuda()
    assert feature_until_from_hooks(hook) == "Up_conv3"
    head = hook._hooks[0]._projector
    assert type(head).__name__ == "DenseProjectionHead" and tuple(head._spatial_size) == (10, 10)
    psd = {k: v.detach().cpu().clone() for k, v in head.state_dict().items()}
    with net.set_grad(False):
        with net.set_grad(True, start="Conv5", end="Up_conv3", include_start=False):
            params = [p for p in net.parameters() if p.requires_grad] + list(hook.parameters())
            flat = ddp.FlatParams(params)
            g = torch.Generator().manual_seed(3)
            img, img_tf = torch.rand(bs, 1, size, size, generator=g), torch.rand(bs, 1, size, size, generator=g)
            filenames, partitions, groups = acdc_like_meta(bs)
            tgt = torch.zeros(bs, 1, 1, 1, dtype=torch.long).cuda()
            batch = ((img.cuda(), img_tf.cuda(), tgt, tgt), filenames, (partitions, groups))
            ep = PretrainDecoderEpocher(model=net, optimizer=torch.optim.SGD([flat.param], lr=0.0),
                                        chain_dataloader=iter([]), num_batches=1, device="cuda",
                                        inference_until="Up_conv3", flat_params=flat)
            ep.add_hooks([hook()])
            with ep.meters.focus_on(ep.meter_focus):
                loss = ep.step_compute(batch, seed=seed)
    # ---- oracle
    def oracle(img, img_tf):
        flips = O.random_flip_decisions(seed, bs)
        x2 = O.apply_flips(img_tf, flips)
        osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
               for k, v in sd.items()}
        feat = O.unet_forward(torch.cat([img, x2], 0), osd, "Up_conv3", train=True, momentum=0.1)
        opsd = {k: v.clone().requires_grad_(True) for k, v in psd.items()}
        f1 = O.apply_flips(feat[:bs], flips)
        z = O.dense_projector_forward(torch.cat([f1, feat[bs:]], 0), opsd, head_type="mlp", normalize=True,
                                      pool_name="adaptive_avg", spatial_size=(10, 10))
        pts = O.dense_region_points(seed, bs, 10, 10, 5)
        sel = lambda zz: torch.cat([torch.stack([zz[b][:, x, y] for x, y in pts[b]]) for b in range(bs)])  # noqa: E731
        a, b = sel(z[:bs]), sel(z[bs:])
        ref = 0.5 * O.supcon_loss(a, b, list(range(a.shape[0])))["loss"]
        ref.backward()
        return ref, osd, opsd

    def oracle_grads(img, img_tf):
        _, osd_, opsd_ = oracle(img, img_tf)
        out = {k: v.grad.numpy() for k, v in osd_.items() if k.startswith("_Up") and torch.is_tensor(v) and v.grad is not None}
        out.update({"head." + k: v.grad.numpy() for k, v in opsd_.items()})
        return out

    ref, osd, opsd = oracle(img, img_tf)
    np.testing.assert_allclose(float(loss.detach()), float(ref.detach()), rtol=2e-4)
    # (split: what a 2e-6 perturbation of the images does to the oracle's own gradients bounds what can be asked)
    # capped at 5e-2; the tie-independent checks of the two product modes are
    # tests/test_gpu_decoder.py::test_f32_split_and_exact_products_agree_call_by_call / ..._wide_statistics_vs_fp64_oracle_fp32
    slack = min(5e-2, 3.0 * oracle_sensitivity((img, img_tf), oracle_grads)) if f32_products == "split" else 0.0
    rel = lambda u, v: float(np.abs(u - v).max() / max(1e-30, np.abs(v).max()))  # noqa: E731
    checked = 0
    for k, p in net.named_parameters():
        if k.startswith(("_Up5", "_Up_conv5", "_Up4", "_Up_conv4", "_Up3", "_Up_conv3")):
            assert p.grad is not None and osd[k].grad is not None, k
            assert rel(p.grad.cpu().numpy(), osd[k].grad.numpy()) < max(5e-3, slack), (k, rel(p.grad.cpu().numpy(), osd[k].grad.numpy()), slack)
            checked += 1
        elif k.startswith("_Conv"):
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k  # frozen encoder
    assert checked >= 18
    # the head's gradients: single elements carry the fp32 noise of the decoder's BatchNorm backward (7e-3 of the largest
    # entry seen on one element of the first 1x1 weight); the tensor as a whole agrees to 3e-3 (relative L2)
    l2 = lambda u, v: float(np.linalg.norm(u - v) / max(1e-30, np.linalg.norm(v)))  # noqa: E731
    for k, p in head.named_parameters():
        got, want = p.grad.cpu().numpy(), opsd[k].grad.numpy()
        assert l2(got, want) < max(3e-3, slack) and rel(got, want) < max(2e-2, slack), (k, l2(got, want), rel(got, want), slack)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape,hid,out", [((3, 64, 19, 23), 256, 256), ((2, 16, 40, 40), 128, 32), ((1, 32, 7, 5), 64, 0),
                                           ((5, 128, 12, 12), 36, 20)])
def test_pixelwise_mlp_rows_gemm_vs_torch(dt, shape, hid, out):
    """``functional.pixelwise_mlp`` as matrix products over the pixel rows (csrc/rows_mlp.hip, round 6) against torch's own
    conv2d -> leaky_relu -> conv2d in float64 on the same (storage-rounded) feature map: output, feature gradient, both
    weight and bias gradients; row counts that are no multiple of the 128-row tile or of the weight gradient's slabs."""
    import spcl_amd  # noqa
    from spcl_amd import functional as F_
    g = torch.Generator().manual_seed(sum(shape) + hid)
    N, C, H, W = shape
    x = torch.randn(N, C, H, W, generator=g).to(dt)
    w1 = torch.randn(hid, C, 1, 1, generator=g) * 0.2
    b1 = torch.randn(hid, generator=g) * 0.1
    mlp = out > 0
    w2 = torch.randn(out, hid, 1, 1, generator=g) * 0.2 if mlp else None
    b2 = torch.randn(out, generator=g) * 0.1 if mlp else None
    r = torch.randn(N, out if mlp else hid, H, W, generator=g)

    def ref():
        xs = x.double().clone().requires_grad_(True)
        ps = [t.double().clone().requires_grad_(True) for t in (w1, b1) + ((w2, b2) if mlp else ())]
        y = torch.nn.functional.conv2d(xs, ps[0], ps[1])
        if mlp:
            y = torch.nn.functional.conv2d(torch.nn.functional.leaky_relu(y, 0.01), ps[2], ps[3])
        (y * r.double()).sum().backward()
        return y.detach(), xs.grad, [p.grad for p in ps]

    y_ref, dx_ref, dp_ref = ref()
    xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    ps = [t.cuda().requires_grad_(True) for t in (w1, b1) + ((w2, b2) if mlp else ())]
    y = F_.pixelwise_mlp(xg, *ps)
    assert tuple(y.shape) == tuple(y_ref.shape) and y.dtype == torch.float32
    (y * r.cuda()).sum().backward()
    tol = 2e-5
    np.testing.assert_allclose(y.detach().cpu().double().numpy(), y_ref.numpy(), rtol=tol, atol=tol * float(y_ref.abs().max()))
    gtol = tol if dt == torch.float32 else 6e-3  # (the feature gradient is stored in the map's own dtype)
    np.testing.assert_allclose(xg.grad.double().cpu().numpy(), dx_ref.numpy(), rtol=gtol, atol=gtol * float(dx_ref.abs().max()))
    for p, want in zip(ps, dp_ref):
        np.testing.assert_allclose(p.grad.double().cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-5 * float(want.abs().max()) + 1e-6)
    # bit-deterministic: the weight gradient's slabs are folded in index order
    xg2 = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    ps2 = [t.cuda().requires_grad_(True) for t in (w1, b1) + ((w2, b2) if mlp else ())]
    (F_.pixelwise_mlp(xg2, *ps2) * r.cuda()).sum().backward()
    assert torch.equal(xg2.grad, xg.grad) and all(torch.equal(a.grad, b.grad) for a, b in zip(ps, ps2))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape,hid,out,hw", [((3, 64, 19, 23), 256, 256, (10, 10)), ((2, 16, 40, 40), 128, 32, (5, 4)),
                                              ((4, 32, 9, 9), 64, 64, (1, 1)), ((2, 128, 14, 14), 36, 20, (14, 14))])
def test_pooled_hidden_form_equals_pool_of_the_projection(dt, shape, hid, out, hw):
    """``pixelwise_mlp_pooled`` (the second 1x1 convolution applied to the POOLED hidden activation: what DenseProjectionHead runs
    for an mlp head with adaptive average pooling) against ``adaptive_pool2d(pixelwise_mlp(...))`` (the reference's order) and
    against float64 torch: same function, sums associated differently."""
    import spcl_amd  # noqa
    from spcl_amd import functional as F_
    g = torch.Generator().manual_seed(sum(shape) + hid + hw[0])
    N, C, H, W = shape
    x = torch.randn(N, C, H, W, generator=g).to(dt)
    w1, b1 = torch.randn(hid, C, 1, 1, generator=g) * 0.2, torch.randn(hid, generator=g) * 0.1
    w2, b2 = torch.randn(out, hid, 1, 1, generator=g) * 0.2, torch.randn(out, generator=g) * 0.1
    r = torch.randn(N, out, *hw, generator=g)

    xs = x.double().clone().requires_grad_(True)
    ps64 = [t.double().clone().requires_grad_(True) for t in (w1, b1, w2, b2)]
    y64 = torch.nn.functional.conv2d(torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(xs, ps64[0], ps64[1]), 0.01),
                                     ps64[2], ps64[3])
    y64 = torch.nn.functional.adaptive_avg_pool2d(y64, hw)
    (y64 * r.double()).sum().backward()

    res = {}
    for form in ("pooled", "reference order"):
        xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        ps = [t.cuda().requires_grad_(True) for t in (w1, b1, w2, b2)]
        if form == "pooled":
            y = F_.pixelwise_mlp_pooled(xg, *ps, hw)
        else:
            y = F_.adaptive_pool2d(F_.pixelwise_mlp(xg, *ps), hw, "avg")
        (y * r.cuda()).sum().backward()
        res[form] = (y.detach(), xg.grad, [p.grad for p in ps])
    y, dx, dps = res["pooled"]
    # (bf16 maps: the pooled form keeps the hidden activation and its gradient in the map's dtype, 8 mantissa bits)
    tol = 3e-5 if dt == torch.float32 else 6e-3
    np.testing.assert_allclose(y.cpu().double().numpy(), y64.detach().numpy(), rtol=tol, atol=tol * float(y64.detach().abs().max()))
    np.testing.assert_allclose(y.cpu().numpy(), res["reference order"][0].cpu().numpy(), rtol=tol, atol=tol * float(y64.detach().abs().max()))
    gtol = tol if dt == torch.float32 else 1e-2
    np.testing.assert_allclose(dx.double().cpu().numpy(), xs.grad.numpy(), rtol=gtol, atol=gtol * float(xs.grad.abs().max()))
    ptol = (1e-4, 2e-5) if dt == torch.float32 else (2e-2, 1e-2)
    for got, want in zip(dps, [p.grad for p in ps64]):
        np.testing.assert_allclose(got.double().cpu().numpy(), want.numpy(), rtol=ptol[0], atol=ptol[1] * float(want.abs().max()) + 1e-6)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_dense_head_at_full_size_vs_float64(dt):
    """The dense projector at a training step's size (30 x 2 maps of 64 channels at 56 x 56 -- Up_conv4's -- = 188 160 pixel rows,
    hidden and output 256, pooled to 10 x 10): the pooled-hidden form and the plain products + pooling, against torch in float64
    ON THE DEVICE -- hundreds of weight-gradient slabs, every tile shape of the products (the unit tests above stop at 3 200
    rows).  With 48 million hidden pre-activations a few lie within rounding of zero, where LeakyReLU' jumps by a factor 100
    and f32 and f64 may disagree about the side: everything downstream of that derivative (feature gradient, first layer's
    gradients) is compared with float64 on the pixels WITHOUT such a near-tie, and between the two device forms (which share the
    pre-activations bit for bit, hence the decisions) everywhere."""
    import spcl_amd  # noqa
    from spcl_amd import functional as F_
    g = torch.Generator().manual_seed(77)
    N, C, H, W, hid, out, hw = 60, 64, 56, 56, 256, 256, (10, 10)
    x = torch.randn(N, C, H, W, generator=g).to(dt)
    w1, b1 = torch.randn(hid, C, 1, 1, generator=g) * 0.1, torch.randn(hid, generator=g) * 0.1
    w2, b2 = torch.randn(out, hid, 1, 1, generator=g) * 0.1, torch.randn(out, generator=g) * 0.1
    r = torch.randn(N, out, *hw, generator=g)

    xs = x.cuda().double().requires_grad_(True)
    ps64 = [t.cuda().double().requires_grad_(True) for t in (w1, b1, w2, b2)]
    pre64 = torch.nn.functional.conv2d(xs, ps64[0], ps64[1])
    y64 = torch.nn.functional.conv2d(torch.nn.functional.leaky_relu(pre64, 0.01), ps64[2], ps64[3])
    y64 = torch.nn.functional.adaptive_avg_pool2d(y64, hw)
    (y64 * r.cuda().double()).sum().backward()
    clear = (pre64.detach().abs().amin(dim=1, keepdim=True) > 1e-4)  # [N, 1, H, W]: no hidden unit of the pixel near a tie
    assert 0.5 < float(clear.double().mean()) < 1.0
    want = {"y": y64.detach(), "dx": xs.grad, "dw1": ps64[0].grad, "db1": ps64[1].grad, "dw2": ps64[2].grad, "db2": ps64[3].grad}
    del y64, pre64

    def rel(a, b):
        return float((a.double() - b.double()).abs().max() / b.double().abs().max())

    forms = {}
    for form in ("pooled", "reference order"):
        xg = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        ps = [t.cuda().requires_grad_(True) for t in (w1, b1, w2, b2)]
        if form == "pooled":
            y = F_.pixelwise_mlp_pooled(xg, *ps, hw)
        else:
            y = F_.adaptive_pool2d(F_.pixelwise_mlp(xg, *ps), hw, "avg")
        (y * r.cuda()).sum().backward()
        got = {"y": y.detach(), "dx": xg.grad, "dw1": ps[0].grad, "db1": ps[1].grad, "dw2": ps[2].grad, "db2": ps[3].grad}
        forms[form] = got
        # against float64: what does not pass through LeakyReLU' everywhere; the feature gradient on the clear pixels
        lo = form == "pooled" and dt == torch.bfloat16  # (hidden activation and its gradient kept in bf16)
        assert rel(got["y"], want["y"]) <= (6e-3 if lo else 3e-5), (form, rel(got["y"], want["y"]))
        assert rel(got["dw2"], want["dw2"]) <= (1e-2 if lo else 2e-4) and rel(got["db2"], want["db2"]) <= (1e-2 if lo else 2e-4), form
        dx_tol = 3e-5 if dt == torch.float32 else 1e-2
        assert rel(got["dx"] * clear, want["dx"] * clear) <= dx_tol, (form, rel(got["dx"] * clear, want["dx"] * clear))
        # (a handful of one-sided decisions among 188 160 x 256 terms: the sums stay close)
        assert rel(got["dw1"], want["dw1"]) <= 2e-2 and rel(got["db1"], want["db1"]) <= 2e-2, form
    a, b = forms["pooled"], forms["reference order"]
    for k in ("dx", "dw1", "db1", "dw2", "db2", "y"):
        assert rel(a[k], b[k]) <= (2e-5 if dt == torch.float32 else 2e-2), (k, rel(a[k], b[k]))
