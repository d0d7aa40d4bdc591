"""GPU parity of the HIP encoder (NHWC MFMA convolutions, BatchNorm statistics, BN-ReLU-pool glue, dgrad, wgrad)
against the golden vectors written from the reference UNet and against the CPU oracle.
fp32 mode: rtol 1e-4-class tolerances (summation order differs); bf16 mode: rtol 2e-2-class."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import spcl_oracle as O


def _unet(max_channel, seed, dtype=torch.float32, input_dim=1):
    import spcl_amd  # noqa
    from spcl_amd.semi_seg.arch import UNet
    m = UNet(input_dim=input_dim, num_classes=4, max_channel=max_channel)
    sd = O.init_unet_state(input_dim, 4, max_channel, seed=seed)
    m.load_state_dict(sd, strict=True)
    m.cuda().train()
    m.set_compute_dtype(dtype)
    return m, sd


def _relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(1e-12, np.abs(b).max())


def test_encoder_golden_small_fp32(golden):
    g = golden("g3_encoder.npz")
    x = torch.tensor(g["small/x"]).cuda()
    for until in ("Conv1", "Conv2", "Conv3", "Conv4"):
        m, _ = _unet(128, 11)
        y = m(x, until=until)
        assert y.shape == g[f"small/out/{until}"].shape
        np.testing.assert_allclose(y.detach().float().cpu().numpy(), g[f"small/out/{until}"], rtol=1e-3, atol=2e-5, err_msg=until)
    m, _ = _unet(128, 11)
    y = m(x, until="Conv5")
    np.testing.assert_allclose(y.detach().cpu().numpy(), g["small/out/Conv5"], rtol=1e-3, atol=2e-5)
    (y * torch.tensor(g["small/r"]).cuda()).sum().backward()
    params = dict(m.named_parameters())
    for k in g.files:
        if k.startswith("small/grad/"):
            name = k[len("small/grad/"):]
            assert params[name].grad is not None, name
            assert _relerr(params[name].grad.cpu().numpy(), g[k]) < 2e-3, (name, _relerr(params[name].grad.cpu().numpy(), g[k]))
        if k.startswith("small/buf/"):
            name = k[len("small/buf/"):]
            buf = dict(m.named_buffers())[name]
            np.testing.assert_allclose(buf.cpu().numpy(), g[k], rtol=1e-4, atol=1e-6, err_msg=name)
    # decoder parameters never received a gradient
    assert params["_Up5.up.1.weight"].grad is None
    m.eval()
    with torch.no_grad():
        ye = m(x, until="Conv5")
    np.testing.assert_allclose(ye.cpu().numpy(), g["small/eval_out/Conv5"], rtol=1e-3, atol=2e-5)
    with pytest.raises(KeyError):
        m(x, until="Conv9")


def test_encoder_golden_base_fp32(golden):
    g = golden("g3_encoder.npz")
    m, _ = _unet(256, 21)
    x = torch.rand(2, 1, 224, 224, generator=torch.Generator().manual_seed(22)).cuda()
    with torch.no_grad():
        y = m(x, until="Conv5")
    assert tuple(y.shape) == (2, 256, 14, 14)
    np.testing.assert_allclose(y.mean(dim=(0, 2, 3)).cpu().numpy(), g["base/out_mean_c"], rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(y[0].cpu().numpy(), g["base/out_n0"], rtol=5e-3, atol=2e-4)


def test_encoder_bf16_drift_vs_fp32_golden(golden):
    """bf16 storage vs the fp32 reference: drift is bounded (it grows ~1.8x per block through batch-stat BN + ReLU
    gating: measured 0.8 % after Conv1 -> 7 % after Conv5 in relative L2); per-channel means stay within 5 %."""
    g = golden("g3_encoder.npz")
    m, _ = _unet(256, 21, torch.bfloat16)
    x = torch.rand(2, 1, 224, 224, generator=torch.Generator().manual_seed(22)).cuda()
    with torch.no_grad():
        y = m(x, until="Conv5")
    assert y.dtype == torch.bfloat16
    ref = g["base/out_n0"]
    err = np.abs(y[0].float().cpu().numpy() - ref)
    assert np.linalg.norm(err) / np.linalg.norm(ref) < 0.15
    np.testing.assert_allclose(y.float().mean(dim=(0, 2, 3)).cpu().numpy(), g["base/out_mean_c"], rtol=8e-2, atol=2e-2)


@pytest.mark.parametrize("shape,mc", [((3, 1, 28, 28), 128), ((2, 1, 48, 40), 128), ((2, 2, 32, 32), 128),
                                      ((2, 1, 112, 112), 256)])
def test_encoder_fp32_vs_oracle_shapes(shape, mc):
    """Tile-edge handling (H,W not multiples of the 14/16 tiles, odd sizes before a pool), multi-channel image input:
    outputs and every encoder parameter gradient of loss = sum(out * r) against the fp32 oracle."""
    n, cin, h, w = shape
    m, sd = _unet(mc, 7, torch.float32, input_dim=cin)
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(*shape, generator=gen)
    sdo = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
           for k, v in sd.items()}
    yr = O.encoder_forward(x, sdo, "Conv5")
    r = torch.randn(yr.shape, generator=gen)
    (yr * r).sum().backward()
    y = m(x.cuda(), until="Conv5")
    assert _relerr(y.detach().cpu().numpy(), yr.detach().numpy()) < 2e-3
    (y * r.cuda()).sum().backward()
    for name, p in m.named_parameters():
        if name.startswith("_Conv"):
            err = _relerr(p.grad.cpu().numpy(), sdo[name].grad.numpy())
            assert err < 5e-3, (name, err)


@pytest.mark.parametrize("shape,mc", [((4, 1, 56, 56), 256), ((2, 2, 32, 32), 128), ((3, 1, 28, 42), 128),
                                      # not multiples of the 14-column conv tiles: shifted last tiles, masked statistics
                                      ((2, 1, 64, 64), 128), ((1, 1, 256, 128), 256), ((3, 1, 100, 72), 128)])
def test_block_bf16_vs_bf16_emulating_oracle(shape, mc):
    """bf16 mode is pinned per block against the oracle run with the SAME storage roundings (oracle.BF16Emulation:
    bf16 weights, raw conv outputs, staged activations and their gradients; fp32 arithmetic).  What is left is
    accumulation order, which moves a few % of the values across a bf16 rounding boundary (1 ulp = 0.4 %):
    measured 1e-3 relative L2 on the block output; tolerance 5e-3 (output) / 6e-2 (gradients of the random-weighted sum, which is dominated by ReLU-gate flips)."""
    n, cin, h, w = shape
    m, sd = _unet(mc, 7, torch.bfloat16, input_dim=cin)
    gen = torch.Generator().manual_seed(5)
    x = torch.rand(*shape, generator=gen)
    sdo = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
           for k, v in sd.items()}
    yr = O.encoder_forward(x, sdo, "Conv1", q=O.BF16Emulation)
    r = torch.randn(yr.shape, generator=gen)
    (yr * r).sum().backward()
    y = m(x.cuda(), until="Conv1")
    rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / max(1e-30, np.linalg.norm(b)))  # noqa: E731
    assert rel(y.detach().float().cpu().numpy(), yr.detach().numpy()) < 5e-3
    (y.float() * r.cuda()).sum().backward()
    for name, p in m.named_parameters():
        if name.startswith("_Conv1"):
            e = rel(p.grad.cpu().numpy(), sdo[name].grad.numpy())
            assert e < 6e-2, (name, e)


def test_encoder_bf16_network_drift_is_that_of_the_emulation():
    """Through the 5 blocks a 1e-3 difference is amplified ~2x per block by batch-stat BN + ReLU gating (a property of
    the network, seen identically between the emulating oracle and the fp32 oracle): the HIP bf16 features stay within
    the same distance of the emulating oracle as that oracle is of fp32 (measured 3.5 % vs 7 % at Conv5)."""
    shape, mc = (2, 1, 112, 112), 256
    m, sd = _unet(mc, 21, torch.bfloat16)
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(22))
    with torch.no_grad():
        y = m(x.cuda(), until="Conv5").float().cpu()
        ye = O.unet_forward(x, {k: v.clone() for k, v in sd.items()}, "Conv5", q=O.BF16Emulation)
        yf = O.unet_forward(x, {k: v.clone() for k, v in sd.items()}, "Conv5")
    d_emu = float((y - ye).norm() / ye.norm())
    d_ref = float((ye - yf).norm() / yf.norm())
    assert d_emu < 0.08 and d_emu < 1.5 * d_ref + 0.01, (d_emu, d_ref)


def test_g4_full_pretrain_step_fp32(golden):
    """One whole pre-train step (encoder -> forward-hook tap -> projector -> self-paced loss -> backward) against the
    reference's own modules (tests/golden/g4_step.npz)."""
    import spcl_amd  # noqa
    from spcl_amd.semi_seg.arch import SingleFeatureExtractor
    from spcl_amd.contrastyou.projectors.heads import ProjectionHead
    from spcl_amd.contrastyou.losses.contrast_loss3 import SelfPacedSupConLoss
    g = golden("g4_step.npz")
    cmax, hid, od, s1, s2 = [int(v) for v in g["dims"]]
    net, _ = _unet(cmax, s1)
    head = ProjectionHead(input_dim=cmax, hidden_dim=hid, output_dim=od, head_type="mlp", normalize=True)
    head.load_state_dict(O.init_projector_state(cmax, hid, od, seed=s2))
    head.cuda()
    crit = SelfPacedSupConLoss(weight_update="soft", correct_grad=True)
    crit.set_gamma(10.0)
    img, img_tf = torch.tensor(g["img"]).cuda(), torch.tensor(g["img_tf"]).cuda()
    n = img.shape[0]
    ext = SingleFeatureExtractor(net, "Conv5")
    ext.bind()
    ext.clear()
    ext.set_enable(True)
    with net.set_grad(False, start="Conv5", include_start=False):
        net(torch.cat([img, img_tf], 0), until="Conv5")
        ext.set_enable(False)
        feat = ext.feature()[-2 * n:]
        z = head(feat)
        a, b = torch.chunk(z, 2)
        loss = crit(a, b, target=g["labels"].tolist())
        loss.backward()
    ext.remove()
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-4)
    np.testing.assert_allclose(crit.downgrade_ratio, g["rho"], rtol=1e-4)
    np.testing.assert_allclose(z.detach().cpu().numpy(), g["z"], rtol=1e-3, atol=1e-5)
    params = dict(net.named_parameters())
    params.update({"proj." + k: p for k, p in head.named_parameters()})
    for k in g.files:
        if k.startswith("grad/"):
            name = k[5:]
            err = _relerr(params[name].grad.cpu().numpy(), g[k])
            assert err < 5e-3, (name, err)


def test_gradient_sinks_fill_the_flat_bucket_in_place():
    """ddp.FlatParams arms a sink per parameter; the backward kernels write the gradients straight into the flat bucket
    (param.grad aliases its slice, gather copies nothing) and the values are bit-identical to the allocate-and-copy path,
    and the bucket also ends up right when a parameter is used twice in one step."""
    import spcl_amd  # noqa
    from spcl_amd import ddp
    from spcl_amd.contrastyou.projectors.heads import ProjectionHead

    def run(sinks, twice=False):
        net, _ = _unet(128, 3)
        head = ProjectionHead(input_dim=128, hidden_dim=32, output_dim=16, head_type="mlp", normalize=True)
        head.load_state_dict(O.init_projector_state(128, 32, 16, seed=5))
        head.cuda()
        params = [p for p in list(net.parameters()) + list(head.parameters())]
        flat = ddp.FlatParams(params, allow_missing_grads=True)  # the whole UNet: the decoder gets no gradient here
        if sinks:
            flat.zero_grad()
        x = torch.rand(4, 1, 32, 32, generator=torch.Generator().manual_seed(1)).cuda()
        with net.set_grad(False, start="Conv5", include_start=False):
            z = head(net(x, until="Conv5"))
            loss = (z * torch.arange(16, device="cuda")).sum()
            if twice:
                loss = loss + 0.5 * head(net(x.flip(3), until="Conv5")).sum()
            loss.backward()
        named = dict(net.named_parameters())
        named.update({"proj." + k: p for k, p in head.named_parameters()})
        direct = {k for k, p in named.items() if p.grad is not None}
        flat.gather_grads()  # (also finishes the conv weights' deferred final sums: they reach their slices here)
        aliased = {k: p.grad is not None and any(p.grad.data_ptr() == v.data_ptr() for v in flat.views)
                   for k, p in named.items()}
        grads = {k: p.grad.clone() for k, p in named.items() if p.grad is not None}
        return grads, aliased, flat.flat.clone(), direct

    g0, a0, f0, d0 = run(False)
    g1, a1, f1, d1 = run(True)
    assert not any(a0.values())  # without sinks autograd allocates and gather copies
    assert len(d0) >= 34  # 30 encoder + 4 projector tensors
    assert len(g1) == len(g0) == len(d0)
    assert all(a1[k] for k in g1), [k for k in g1 if not a1[k]]
    # with sinks autograd never sees the conv weights: their final sums ride in the batched reduction at gather time
    late = set(g1) - d1
    assert late and all(k.endswith("weight") and g1[k].dim() == 4 for k in late), late
    for k in g0:
        if k in late:  # same partial sums, another summation tree
            assert _relerr(g1[k].cpu().numpy(), g0[k].cpu().numpy()) < 2e-6, k
        else:
            assert torch.equal(g0[k], g1[k]), k
    assert _relerr(f1.cpu().numpy(), f0.cpu().numpy()) < 2e-6
    g2, _, f2, _ = run(False, twice=True)
    g3, _, f3, _ = run(True, twice=True)  # autograd sums the two uses itself; the bucket gets the sum either way
    for k in g2:
        assert _relerr(g3[k].cpu().numpy(), g2[k].cpu().numpy()) < 1e-6, k
    assert _relerr(f3.cpu().numpy(), f2.cpu().numpy()) < 1e-6


def test_deferred_wide_weight_gradients_bf16():
    """bf16: the weight gradients of the >=64-channel layers are queued during backward and computed by ONE batched launch
    when the flat bucket is gathered (functional.DeferredWgrads).  Same values as the immediate per-layer path (other
    pixel splits -> fp32 summation order only), also when a parameter is used twice in the step (the second use does not
    get the sink; autograd adds it into the bucket slice BEFORE the flush, which therefore adds instead of overwriting),
    and a forward without backward leaves no armed sink behind."""
    import spcl_amd  # noqa
    from spcl_amd import ddp, functional as F_
    from spcl_amd.contrastyou.projectors.heads import ProjectionHead

    def run(sinks, twice):
        net, _ = _unet(256, 3, torch.bfloat16)
        head = ProjectionHead(input_dim=256, hidden_dim=32, output_dim=16, head_type="mlp", normalize=True)
        head.load_state_dict(O.init_projector_state(256, 32, 16, seed=5))
        head.cuda()
        for name in net.decoder_names:
            getattr(net, "_" + name).requires_grad_(False)
        params = [p for p in list(net.parameters()) + list(head.parameters()) if p.requires_grad]
        flat = ddp.FlatParams(params)
        x = torch.rand(4, 1, 64, 64, generator=torch.Generator().manual_seed(1)).cuda()
        with torch.no_grad():
            net(x, until="Conv5")  # a forward that never runs backward
        z = head(net(x, until="Conv5"))
        loss = (z * torch.arange(16, device="cuda")).sum()
        if twice:
            loss = loss + 0.5 * head(net(x.flip(3), until="Conv5")).sum()
        if sinks:
            flat.zero_grad()
            assert flat._queue is not None and all(F_.sink_queue(v) is flat._queue for v in flat.views)
        loss.backward()
        queued = len(flat._queue.items) if getattr(flat, "_queue", None) is not None else 0
        flat.gather_grads()
        assert flat._queue is None and not any(getattr(p, "_grad_sink_armed", False) for p in params)
        assert all(F_.sink_queue(v) is None for v in flat.views)
        return flat.flat.clone(), queued, {k: p.grad for k, p in net.named_parameters() if p.grad is not None}

    for twice in (False, True):
        f0, q0, _ = run(False, twice)
        f1, q1, g1 = run(True, twice)
        assert q0 == 0 and q1 == 5  # _Conv3.b, _Conv4.a/b, _Conv5.a/b (64..256 channels)
        assert _relerr(f1.cpu().numpy(), f0.cpu().numpy()) < 2e-3, twice
        w = g1["_Conv5.conv.3.weight"]
        assert w.abs().max() > 0 and torch.isfinite(w).all()


@pytest.mark.parametrize("N,H,W", [(3, 224, 224), (2, 140, 154), (5, 448, 224)])
def test_image_conv_leaves_the_autocorrelation_rows(N, H, W):
    """spcl_conv3x3_forward_image_acorr (csrc/conv_fast.hip conv3x3_image_kernel<.., ACORR>): the first convolution of the
    image block (unet.py:123, 1 -> 16 channels) that also leaves the image's autocorrelation partial rows, one per 14 x 14
    tile -- y and the BatchNorm partials bit for bit those of spcl_conv3x3_forward (in_mode 2), the rows' totals those of the
    stand-alone pass (spcl_image_autocorr: other partial sums, f32 order) and of numpy on the bf16-rounded, zero-padded
    image; twice (fixed-order sums).  Sizes the tile does not divide are refused (the rows come from their own pass)."""
    import spcl_amd  # noqa
    from spcl_amd import functional as F_, native as _n
    dtc = _n.dtype_code(torch.bfloat16)
    g = torch.Generator().manual_seed(7 + N)
    img = torch.rand(N, H, W, generator=g).cuda()
    w = torch.randn(16, 1, 3, 3, generator=g).cuda() * 0.3
    wp = F_._pack(w, 0, dtc, torch.bfloat16)
    rows = F_._acorr_in_conv_rows(dtc, N, H, W)
    assert rows == N * (H // 14) * (W // 14)
    assert F_._acorr_in_conv_rows(dtc, 1, 256, 256) == 0 and F_._acorr_in_conv_rows(dtc, 1, 112, 112) == 0
    y0, s0 = F_._conv(img.view(N, H, W, 1), dtc, torch.bfloat16, N, H, W, 1, 16, 16, wp, 2, None, None, True)
    y1, s1, ac = F_._conv_image_acorr(img.view(N, H, W, 1), dtc, torch.bfloat16, N, H, W, 1, 16, wp, True, rows)
    y2, s2, ac2 = F_._conv_image_acorr(img.view(N, H, W, 1), dtc, torch.bfloat16, N, H, W, 1, 16, wp, True, rows)
    assert torch.equal(y0, y1) and torch.equal(s0[:s0.ntiles * 48], s1[:s1.ntiles * 48])
    assert torch.equal(ac, ac2) and torch.equal(y1, y2)
    tot = ac.double().sum(0).cpu().numpy()
    ref = F_._image_autocorr(img.contiguous(), N, H, W).double().sum(0).cpu().numpy()
    assert _relerr(tot[:54], ref[:54]) < 1e-6 and np.abs(tot[54:]).max() == 0.0
    a = torch.nn.functional.pad(img.bfloat16().double().cpu(), (2, 2, 2, 2))
    shift = lambda k: a[:, 1 + k // 3:1 + k // 3 + H, 1 + k % 3:1 + k % 3 + W]  # noqa: E731  img0[p + tap - 1]
    k = 0
    for u in range(9):
        for v in range(u, 9):
            want = float((shift(u) * shift(v)).sum())
            assert abs(tot[k] - want) < 2e-6 * abs(want) + 1e-6, (u, v, tot[k], want)
            k += 1
    for t in range(9):
        want = float(shift(t).sum())
        assert abs(tot[45 + t] - want) < 2e-6 * abs(want) + 1e-6


def test_image3_first_layer_gradient_without_a_pass_over_its_output():
    """bf16, one-channel image, 224 x 224 (14 x 14 tiles): the first conv's weight gradient and BN backward come from the
    Conv1.b dgrad epilogue's tap sums + the image autocorrelation (csrc/bn.hip image3: dW = scale S1 + A (W R) + B S3) instead
    of the fused pass over y and g.  Same values as that pass (which tests/test_gpu_kernels.py holds against fp64 math):
    dgamma / dbeta to f32 summation order, dW within bf16 storage noise; and the standalone kernels against numpy."""
    import spcl_amd  # noqa
    from spcl_amd import functional as F_, native as _n
    g = torch.Generator().manual_seed(2)
    img = torch.rand(3, 1, 224, 224, generator=g).cuda()
    # ---- the autocorrelation kernel alone
    acorr = F_._image_autocorr(img.permute(0, 2, 3, 1).contiguous(), 3, 224, 224).double().sum(0).cpu().numpy()
    a = torch.nn.functional.pad(img.bfloat16().double().cpu(), (2, 2, 2, 2))[:, 0]
    f = torch.nn.functional.pad(img.double().cpu(), (2, 2, 2, 2))[:, 0]
    shift = lambda t, k: t[:, 1 + k // 3:1 + k // 3 + 224, 1 + k % 3:1 + k % 3 + 224]  # noqa: E731  img0[p + tap - 1]
    k = 0
    for u in range(9):
        for v in range(u, 9):
            want = float((shift(a, u) * shift(a, v)).sum())
            assert abs(acorr[k] - want) < 2e-6 * abs(want), (u, v, acorr[k], want)
            k += 1
    for t in range(9):  # the nine image sums ride on the matrix pipe too: the bf16-rounded image, zero-mean rounding errors
        want, exact = float(shift(a, t).sum()), float(shift(f, t).sum())
        assert abs(acorr[45 + t] - want) < 2e-6 * abs(want), (t, acorr[45 + t], want)
        assert abs(acorr[45 + t] - exact) < 1e-4 * abs(exact), (t, acorr[45 + t], exact)
    # ---- the dgrad entry with and without the gradient tensor: same rows, and g is the plain dgrad
    gq = torch.Generator().manual_seed(4)
    dyq = torch.randn(2, 140, 140, 16, generator=gq).cuda().bfloat16()
    y2q = torch.randn(2, 140, 140, 16, generator=gq).cuda().bfloat16()
    wq = torch.randn(16, 16, 3, 3, generator=gq).cuda() * 0.1
    stq = [torch.randn(16, generator=gq).cuda() * 0.1, torch.rand(16, generator=gq).cuda() + 0.5,
           torch.rand(16, generator=gq).cuda() + 0.5, torch.randn(16, generator=gq).cuda() * 0.1]
    imq = torch.rand(2, 140, 140, generator=gq).cuda()
    wpt = F_._pack(wq, 1, _n.dtype_code(torch.bfloat16), torch.bfloat16)
    g_a, rows_a = F_._dgrad_bnstats_image(dyq, wpt, y2q, stq, imq, _n.dtype_code(torch.bfloat16), torch.bfloat16, 2, 140, 140,
                                          16, want_g=True)
    g_b, rows_b = F_._dgrad_bnstats_image(dyq, wpt, y2q, stq, imq, _n.dtype_code(torch.bfloat16), torch.bfloat16, 2, 140, 140,
                                          16)
    assert g_b is None and torch.equal(rows_a, rows_b) and rows_a.abs().max() > 0
    want_g = torch.nn.functional.conv_transpose2d(dyq.float().permute(0, 3, 1, 2), wq.bfloat16().float(), padding=1)
    assert _relerr(g_a.float().permute(0, 3, 1, 2).cpu().numpy(), want_g.cpu().numpy()) < 6e-3
    # ---- the whole block, new path against the old one
    res, default = {}, F_._IMAGE3
    for on in (False, True, True):
        F_._IMAGE3 = on
        try:
            net, _ = _unet(256, 7, torch.bfloat16)
            for name in net.decoder_names:
                getattr(net, "_" + name).requires_grad_(False)
            out = net(img, until="Conv2")
            w = torch.rand(out.shape, generator=torch.Generator().manual_seed(3)).cuda().to(out.dtype)
            (out.float() * w.float()).sum().backward()
            c1 = net._Conv1.conv
            grads = [p.grad.detach().float().cpu().clone() for p in (c1[0].weight, c1[1].weight, c1[1].bias, c1[3].weight)]
            if on in res:  # the second run of the new path: bit-identical (fixed-order sums, no racing reads)
                assert all(torch.equal(a, b) for a, b in zip(res[on], grads))
            res[on] = grads
        finally:
            F_._IMAGE3 = default
    dw0, dg0, db0, dwb0 = res[False]
    dw1, dg1, db1, dwb1 = res[True]
    # dgamma / dbeta: the same per-pixel terms; the one-pass kernel adds two waves' shares per tile (another f32 order)
    assert _relerr(dg1.numpy(), dg0.numpy()) < 2e-6 and _relerr(db1.numpy(), db0.numpy()) < 2e-6
    assert _relerr(dwb1.numpy(), dwb0.numpy()) < 2e-5  # (the one-pass kernel sums the same products in another order)
    assert dw0.abs().max() > 0
    assert _relerr(dw1.numpy(), dw0.numpy()) < 3e-3, _relerr(dw1.numpy(), dw0.numpy())


@pytest.mark.parametrize("N,H,W", [(2, 140, 154), (9, 224, 224), (1, 256, 256), (3, 140, 14)])
def test_conv16_backward_in_one_pass(N, H, W):
    """csrc/conv16_bwd.hip: the image block's second conv (unet.py:75, 16 -> 16 channels) -- weight gradient and the rows the
    first conv's backward is finished from, in one launch.  Rows: the dgrad kernel's up to the order of the f32 sums;
    dW: against fp64 math on the same bf16 operands, and next to the stand-alone weight-gradient kernel.  256 x 256 has
    shifted last tiles (pixels two tiles cover count once); N = 9 at 224 x 224 makes workgroups walk two images (the last one);
    140 x 14 has 10 tiles per image, fewer than the 16 folded autocorrelation rows its extra grid slice must write."""
    import spcl_amd  # noqa
    from spcl_amd import functional as F_, native as _n
    dtc = _n.dtype_code(torch.bfloat16)
    gq = torch.Generator().manual_seed(11 + N)
    dy = torch.randn(N, H, W, 16, generator=gq).cuda().bfloat16()
    y2 = torch.randn(N, H, W, 16, generator=gq).cuda().bfloat16()
    w = torch.randn(16, 16, 3, 3, generator=gq).cuda() * 0.1
    st = [torch.randn(16, generator=gq).cuda() * 0.1, torch.rand(16, generator=gq).cuda() + 0.5,
          torch.rand(16, generator=gq).cuda() + 0.5, torch.randn(16, generator=gq).cuda() * 0.3]  # mean, -, scale, shift
    img = torch.rand(N, H, W, generator=gq).cuda()
    assert _n.call("spcl_conv16_bwd_fused_supported", dtc, N, H, W, 16, 16)
    wpt = F_._pack(w, 1, dtc, torch.bfloat16)
    _, rows_ref = F_._dgrad_bnstats_image(dy, wpt, y2, st, img, dtc, torch.bfloat16, N, H, W, 16)
    dw_old = F_._wgrad(y2, dy, dtc, N, H, W, 16, 16, 16, 16, 16, 1, st[2], st[3], None)
    for _ in range(2):
        dw, rows = F_._conv16_bwd_fused(dy, wpt, y2, st, img, dtc, N, H, W, 16, 16, 16, None)
        # (the two BatchNorm sums are combined from four waves' shares, the tap sums walk the pixels in another order)
        assert _relerr(rows.cpu().numpy(), rows_ref.cpu().numpy()) < 2e-6
        x = torch.relu(torch.addcmul(st[3], y2.float(), st[2])).bfloat16().double().permute(0, 3, 1, 2)
        want = torch.nn.grad.conv2d_weight(x, (16, 16, 3, 3), dy.double().permute(0, 3, 1, 2), padding=1)
        assert _relerr(dw.cpu().numpy(), want.cpu().numpy()) < 2e-5, _relerr(dw.cpu().numpy(), want.cpu().numpy())
        assert _relerr(dw.cpu().numpy(), dw_old.cpu().numpy()) < 2e-5
        if _ == 0:
            first = (dw.clone(), rows.clone())
    assert torch.equal(first[0], dw) and torch.equal(first[1], rows)  # fixed-order sums
    # ---- one row set per workgroup (its tiles summed in registers) in the final kernel's layout, the autocorrelation's rows
    # folded by the same launch: the same totals, and the final kernel gives the same dgamma / dbeta / dW as from the tiles
    acorr = F_._image_autocorr(img.contiguous(), N, H, W)
    dw_w, rows_w = F_._conv16_bwd_fused(dy, wpt, y2, st, img, dtc, N, H, W, 16, 16, 16, None, acorr=acorr)
    assert torch.equal(dw_w, dw) and rows_w.wg == _n.call("spcl_conv16_bwd_fused_splits", N, H, W)
    tot_w = rows_w.view(11, 16, rows_w.wg).double().sum(2)
    tot_t = rows.view(-1, 11, 16).double().sum(0)
    assert _relerr(tot_w.cpu().numpy(), tot_t.cpu().numpy()) < 1e-6
    assert _relerr(rows_w.acorr16.double().sum(0).cpu().numpy(), acorr.double().sum(0).cpu().numpy()) < 1e-6
    w1 = torch.randn(16, 1, 3, 3, generator=gq).cuda() * 0.3
    st1 = [st[0], torch.rand(16, generator=gq).cuda() + 0.5, st[2], st[3]]  # mean, invstd, scale, shift
    fin_w = F_._bnrelu_bwd_rows_image3(rows_w, acorr, w1, N, H, W, 16, 16, st1, True, (None, None, None))
    fin_t = F_._bnrelu_bwd_rows_image3(rows, acorr, w1, N, H, W, 16, 16, st1, True, (None, None, None))
    for a_, b_ in zip(fin_w, fin_t):
        assert _relerr(a_.cpu().numpy(), b_.cpu().numpy()) < 1e-5 and b_.abs().max() > 0


def test_two_buckets_armed_in_one_step_keep_their_own_deferred_gradients():
    """the deferred weight-gradient queue belongs to the bucket that armed the sink.  Two buckets armed in one
    step (Conv1..Conv4 | Conv5 + head) both receive their wide layers' gradients whatever the gather order, and a step
    aborted after arming leaves nothing queued for the next one."""
    import spcl_amd  # noqa
    from spcl_amd import ddp
    from spcl_amd.contrastyou.projectors.heads import ProjectionHead

    def run(two_buckets, order=(0, 1), abort_first=False):
        net, _ = _unet(256, 3, torch.bfloat16)
        head = ProjectionHead(input_dim=256, hidden_dim=32, output_dim=16, head_type="mlp", normalize=True)
        head.load_state_dict(O.init_projector_state(256, 32, 16, seed=5))
        head.cuda()
        for name in net.decoder_names:
            getattr(net, "_" + name).requires_grad_(False)
        pa = [p for k, p in net.named_parameters() if p.requires_grad and not k.startswith("_Conv5")]
        pb = [p for k, p in net.named_parameters() if p.requires_grad and k.startswith("_Conv5")] + list(head.parameters())
        buckets = [ddp.GradBucket(pa), ddp.GradBucket(pb)] if two_buckets else [ddp.GradBucket(pa + pb)]
        x = torch.rand(4, 1, 64, 64, generator=torch.Generator().manual_seed(1)).cuda()

        def fwd():
            z = head(net(x, until="Conv5"))
            return (z * torch.arange(16, device="cuda")).sum()
        if abort_first:  # arm, run half a step, never gather
            for b in buckets:
                b.arm_sinks()
            fwd().backward()
            for p in pa + pb:
                p.grad = None
        for b in buckets:
            b.arm_sinks()
        fwd().backward()
        for i in (order if two_buckets else (0,)):
            buckets[i].gather()
        return torch.cat([b.flat for b in buckets]).clone()

    ref = run(False)
    for order in ((0, 1), (1, 0)):
        got = run(True, order)
        assert torch.equal(got, ref), order
    assert torch.equal(run(True, (0, 1), abort_first=True), ref)
    assert ref.abs().max() > 0


def test_three_combined_hooks_share_one_encoder_pass_fp32():
    """SURVEY row N4 / BASELINE configs[3] shape: three self-paced hooks on Conv5 (partition, patient, self meta-labels,
    weights 1 / 0.5 / 0.25, each with its own projector) through the pre-train epocher's step_compute, against the oracle:
    ONE encoder forward, three projector + loss evaluations, weighted sum, one backward."""
    import spcl_amd  # noqa
    from spcl_amd import ddp
    from spcl_amd.semi_seg.epochers import PretrainEncoderEpocher
    from spcl_amd.semi_seg.hooks import create_sp_infonce_hooks
    from spcl_amd.synthetic import prostate_like_meta
    net, sd = _unet(128, 11)
    ons, weights, bs = ["partition", "patient", "self"], [1.0, 0.5, 0.25], 16
    hook = create_sp_infonce_hooks(model=net, feature_names=["Conv5"] * 3, weights=weights, contrast_ons=ons,
                                   begin_values=8.0, end_values=8.0, mode="soft", max_epoch=10, p=0.5, correct_grad=True,
                                   data_name="prostate", sync_checks=True).cuda()
    for name in net.decoder_names:
        getattr(net, "_" + name).requires_grad_(False)
    params = [p for p in net.parameters() if p.requires_grad] + list(hook.parameters())
    flat = ddp.FlatParams(params)
    g = torch.Generator().manual_seed(5)
    img, img_tf = torch.rand(bs, 1, 32, 32, generator=g), torch.rand(bs, 1, 32, 32, generator=g)
    filenames, partitions, groups = prostate_like_meta(bs, partition_num=4)
    tgt = torch.zeros(bs, 1, 1, 1, dtype=torch.long)
    batch = ((img.cuda(), img_tf.cuda(), tgt.cuda(), tgt.cuda()), filenames, (partitions, groups))
    opt = torch.optim.SGD([flat.param], lr=0.0)
    ep = PretrainEncoderEpocher(model=net, optimizer=opt, chain_dataloader=iter([batch]), num_batches=1, device="cuda",
                                inference_until="Conv5", flat_params=flat)
    # projector weights BEFORE the step, for the oracle
    heads = [{k: v.detach().cpu().clone() for k, v in h._projector.state_dict().items()} for h in hook._hooks]
    ep.add_hooks([hook()])
    net.train()
    with ep.meters.focus_on(ep.meter_focus):
        loss = ep.step_compute(batch, seed=3)
    # ---- oracle: view 2 is flipped by the epocher with the seeded per-sample flips
    flips = O.random_flip_decisions(3, bs)
    x2 = O.apply_flips(img_tf, flips)
    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
           for k, v in sd.items()}
    feat = O.encoder_forward(torch.cat([img, x2], 0), osd, "Conv5", train=True, momentum=0.1)
    total, leaves = 0.0, {}
    for hi, (on, w, psd) in enumerate(zip(ons, weights, heads)):
        psd = {k: v.clone().requires_grad_(True) for k, v in psd.items()}
        z = O.projector_forward(feat, psd)
        labels = O.get_label(on, "prostate", partitions, groups)
        r = O.supcon_loss(z[:bs], z[bs:], labels, gamma=8.0, mode="soft", correct_grad=True)
        total = total + w * r["loss"]
        leaves.update({f"h{hi}.{k}": v for k, v in psd.items()})
    total.backward()
    np.testing.assert_allclose(loss.item(), float(total.detach()), rtol=2e-4)
    named = dict(net.named_parameters())
    for k, p in named.items():
        if p.requires_grad and osd[k].grad is not None:
            assert _relerr(p.grad.cpu().numpy(), osd[k].grad.numpy()) < 5e-3, k
    for hi, h in enumerate(hook._hooks):
        for k, p in h._projector.named_parameters():
            assert _relerr(p.grad.cpu().numpy(), leaves[f"h{hi}.{k}"].grad.numpy()) < 5e-3, (hi, k)


def test_pretrain_loss_curve_matches_oracle_over_steps_fp32():
    """Six consecutive pre-train steps (encoder -> tap -> projector -> self-paced loss -> backward -> RAdam on the flat
    parameter, fresh batch and flip seed every step) against the oracle driven by torch.optim.RAdam on CPU: the loss curve
    and the parameters after the last step agree within fp32 tolerance (the north star's "contrastive loss curve")."""
    import spcl_amd  # noqa
    from spcl_amd import ddp
    from spcl_amd.optim import FusedRAdam
    from spcl_amd.semi_seg.epochers import PretrainEncoderEpocher
    from spcl_amd.semi_seg.hooks import create_sp_infonce_hooks
    from spcl_amd.synthetic import acdc_like_meta
    net, sd = _unet(128, 21)
    bs, steps, lr, wd, gamma = 12, 6, 2e-3, 1e-5, 10.0
    hook = create_sp_infonce_hooks(model=net, feature_names="Conv5", weights=1.0, contrast_ons="partition",
                                   begin_values=gamma, end_values=gamma, mode="soft", max_epoch=10, p=0.5,
                                   correct_grad=True, data_name="acdc", sync_checks=True).cuda()
    for name in net.decoder_names:
        getattr(net, "_" + name).requires_grad_(False)
    enc_names = [k for k, p in net.named_parameters() if p.requires_grad]
    head = hook._hooks[0]._projector
    psd0 = {k: v.detach().cpu().clone() for k, v in head.state_dict().items()}
    flat = ddp.FlatParams([p for p in net.parameters() if p.requires_grad] + list(hook.parameters()))
    opt = FusedRAdam([flat.param], lr=lr, weight_decay=wd)
    filenames, partitions, groups = acdc_like_meta(bs)
    g = torch.Generator().manual_seed(17)
    batches = [(torch.rand(bs, 1, 32, 32, generator=g), torch.rand(bs, 1, 32, 32, generator=g)) for _ in range(steps)]
    tgt = torch.zeros(bs, 1, 1, 1, dtype=torch.long).cuda()
    ep = PretrainEncoderEpocher(model=net, optimizer=opt, chain_dataloader=iter([]), num_batches=steps, device="cuda",
                                inference_until="Conv5", flat_params=flat)
    ep.add_hooks([hook()])
    net.train()
    curve = []
    with ep.meters.focus_on(ep.meter_focus):
        for k, (a, b) in enumerate(batches):
            batch = ((a.cuda(), b.cuda(), tgt, tgt), filenames, (partitions, groups))
            curve.append(float(ep.step(batch, seed=100 + k).detach()))
    # ---- oracle
    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
           for k, v in sd.items()}
    opsd = {k: v.clone().requires_grad_(True) for k, v in psd0.items()}
    leaves = [osd[k] for k in enc_names] + [opsd[k] for k, _ in head.named_parameters()]
    oopt = torch.optim.RAdam(leaves, lr=lr, weight_decay=wd)
    labels = O.get_label("partition", "acdc", partitions, groups)
    ocurve = []
    for k, (a, b) in enumerate(batches):
        x2 = O.apply_flips(b, O.random_flip_decisions(100 + k, bs))
        feat = O.encoder_forward(torch.cat([a, x2], 0), osd, "Conv5", train=True, momentum=0.1)
        z = O.projector_forward(feat, opsd)
        r = O.supcon_loss(z[:bs], z[bs:], labels, gamma=gamma, mode="soft", correct_grad=True)
        oopt.zero_grad()
        r["loss"].backward()
        oopt.step()
        ocurve.append(float(r["loss"].detach()))
    np.testing.assert_allclose(curve, ocurve, rtol=2e-3)
    assert abs(curve[-1] - curve[0]) > 1e-3  # the parameters really moved
    named = dict(net.named_parameters())
    for k in enc_names:
        delta = (osd[k].detach() - sd[k]).abs().max().item()
        err = (named[k].detach().cpu() - osd[k].detach()).abs().max().item()
        # RAdam's normalised update moves a parameter by ~lr per step whatever the gradient's size, so a gradient element
        # that is rounding noise on both sides (CPU thread order vs MFMA order) may step the other way: absolute floor
        assert err < 0.1 * delta + 3e-5, (k, err, delta)


@pytest.mark.parametrize("steps,rtol", [(24, 5e-3), (96, 5e-3)])
def test_pretrain_loss_curve_bf16_tracks_the_fp32_oracle_over_many_steps(steps, rtol):
    """The benchmarked dtype over a multi-step run (the north star's "contrastive loss curve"): 24 and 96 consecutive
    pre-train steps with bf16 activation storage (fresh batch, slice order and flip seed every step, FusedRAdam on the flat
    parameter -- its coefficients staged from the host --, the step replayed from the epocher's hipGraph from the third step
    on) next to the fp32 CPU oracle driven by torch.optim.RAdam: every step's loss within 0.5 % (measured: 0.2 % at worst over
    96 steps -- on random slices the loss stays near log(2n - 1), so this is a per-step agreement of two trajectories in different
    arithmetic, not a learning curve), and the curve really moves."""
    import spcl_amd  # noqa
    from spcl_amd import ddp
    from spcl_amd.optim import FusedRAdam
    from spcl_amd.semi_seg.epochers import PretrainEncoderEpocher
    from spcl_amd.semi_seg.hooks import create_sp_infonce_hooks
    from spcl_amd.synthetic import acdc_like_meta
    net, sd = _unet(128, 23)
    net.set_compute_dtype(torch.bfloat16)
    bs, lr, wd, gamma, size = 12, 2e-3, 1e-5, 10.0, 64
    hook = create_sp_infonce_hooks(model=net, feature_names="Conv5", weights=1.0, contrast_ons="partition",
                                   begin_values=gamma, end_values=gamma, mode="soft", max_epoch=10, p=0.5,
                                   correct_grad=True, data_name="acdc", sync_checks=False).cuda()
    for name in net.decoder_names:
        getattr(net, "_" + name).requires_grad_(False)
    enc_names = [k for k, p in net.named_parameters() if p.requires_grad]
    head = hook._hooks[0]._projector
    psd0 = {k: v.detach().cpu().clone() for k, v in head.state_dict().items()}
    flat = ddp.FlatParams([p for p in net.parameters() if p.requires_grad] + list(hook.parameters()))
    opt = FusedRAdam([flat.param], lr=lr, weight_decay=wd)
    g = torch.Generator().manual_seed(19)
    batches = [(torch.rand(bs, 1, size, size, generator=g), torch.rand(bs, 1, size, size, generator=g))
               for _ in range(steps)]
    metas = [acdc_like_meta(bs, shift=5 * k) for k in range(steps)]
    tgt = torch.zeros(bs, 1, 1, 1, dtype=torch.long).cuda()
    ep = PretrainEncoderEpocher(model=net, optimizer=opt, chain_dataloader=iter([]), num_batches=steps, device="cuda",
                                inference_until="Conv5", flat_params=flat)
    ep.add_hooks([hook()])
    net.train()
    curve = []
    with ep.meters.focus_on(ep.meter_focus):
        for k, (a, b) in enumerate(batches):
            fn, part, grp = metas[k]
            curve.append(ep.step(((a.cuda(), b.cuda(), tgt, tgt), fn, (part, grp)), seed=300 + k).detach().clone())
    curve = [float(c) for c in curve]
    assert ep._step_graph is not None and ep._step_graph.captured and ep._step_graph.replays == steps - 2
    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
           for k, v in sd.items()}
    opsd = {k: v.clone().requires_grad_(True) for k, v in psd0.items()}
    leaves = [osd[k] for k in enc_names] + [opsd[k] for k, _ in head.named_parameters()]
    oopt = torch.optim.RAdam(leaves, lr=lr, weight_decay=wd)
    ocurve = []
    for k, (a, b) in enumerate(batches):
        fn, part, grp = metas[k]
        labels = O.get_label("partition", "acdc", part, grp)
        x2 = O.apply_flips(b, O.random_flip_decisions(300 + k, bs))
        feat = O.encoder_forward(torch.cat([a, x2], 0), osd, "Conv5", train=True, momentum=0.1)
        z = O.projector_forward(feat, opsd)
        r = O.supcon_loss(z[:bs], z[bs:], labels, gamma=gamma, mode="soft", correct_grad=True)
        oopt.zero_grad()
        r["loss"].backward()
        oopt.step()
        ocurve.append(float(r["loss"].detach()))
    print("bf16 HIP :", [round(c, 4) for c in curve])
    print("fp32 orac:", [round(c, 4) for c in ocurve])
    np.testing.assert_allclose(curve, ocurve, rtol=rtol)
    assert max(ocurve) - min(ocurve) > 0.02  # the parameters really moved


def test_bn_kat5_statistics():
    """KAT-5: after the first block the fused BN has mean 0 / biased var 1 before the affine; running_var uses the
    unbiased variance with momentum 0.1."""
    m, sd = _unet(128, 5)
    x = torch.rand(4, 1, 16, 16, generator=torch.Generator().manual_seed(6))
    y = torch.nn.functional.conv2d(x, sd["_Conv1.conv.0.weight"], None, 1, 1)
    m(x.cuda(), until="Conv1")
    bn = m._Conv1.conv[1]
    np.testing.assert_allclose(bn.running_mean.cpu().numpy(), 0.1 * y.mean(dim=(0, 2, 3)).numpy(), rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(bn.running_var.cpu().numpy(), 0.9 + 0.1 * y.var(dim=(0, 2, 3), unbiased=True).numpy(),
                               rtol=1e-4)
    assert int(bn.num_batches_tracked) == 1
    with m.set_bn_track(False):
        m(x.cuda(), until="Conv1")
    assert int(bn.num_batches_tracked) == 1


def test_unet_api_surface():
    import spcl_amd  # noqa
    from spcl_amd.semi_seg.arch import UNet, get_channel_dim, sort_arch
    assert get_channel_dim("Conv5") == 256 and get_channel_dim("Conv1", max_channel=128) == 8
    assert sort_arch(["Up_conv3", "Conv2", "Conv5"]) == ["Conv2", "Conv5", "Up_conv3"]
    with pytest.raises(AssertionError):
        UNet(max_channel=100)
    m = UNet(input_dim=1, num_classes=4).cuda()
    assert m.num_classes == 4 and m.get_channel_dim("Deconv_1x1") == 4
    with m.set_grad(False, start="Conv5", include_start=False):
        assert all(p.requires_grad for p in m._Conv5.parameters())
        assert not any(p.requires_grad for p in m._Up5.parameters())
    assert all(p.requires_grad for p in m._Up5.parameters())
    with pytest.raises(ValueError):
        with m.set_grad(False, start=None, include_start=False):
            pass
    with pytest.raises(RuntimeError):  # CPU tensor: no fallback
        m.cpu()(torch.zeros(1, 1, 16, 16), until="Conv1")


def test_two_bucket_overlap_step_equals_the_plain_step():
    """ddp.enable_unet_overlap on one process: the early bucket (Conv3 .. Conv5 + projector) is gathered from the backward
    hook on Conv2 -- the batched wide weight gradients are flushed there, the narrow layers' sinks are still being
    written -- and the rest after backward: the flat gradient is bit-identical to the one-gather step's."""
    import spcl_amd  # noqa
    from spcl_amd import ddp
    from spcl_amd.semi_seg.epochers import PretrainEncoderEpocher
    from spcl_amd.semi_seg.hooks import create_sp_infonce_hooks
    from spcl_amd.synthetic import acdc_like_meta
    grads = []
    for overlap in (False, True):
        torch.manual_seed(9)  # the projector head draws its initial weights from the global generator
        net, _ = _unet(256, 21)
        net.set_compute_dtype(torch.bfloat16)
        hook = create_sp_infonce_hooks(model=net, feature_names="Conv5", weights=1.0, contrast_ons="partition",
                                       begin_values=8.0, end_values=8.0, mode="soft", max_epoch=10, p=0.5,
                                       correct_grad=True, data_name="acdc", sync_checks=False).cuda()
        for name in net.decoder_names:
            getattr(net, "_" + name).requires_grad_(False)
        flat = ddp.FlatParams([p for p in net.parameters() if p.requires_grad] + list(hook.parameters()))
        fired = []
        if overlap:
            ddp.enable_unet_overlap(flat, net)
            start = net._boundary_hooks["Conv2"]
            net._boundary_hooks["Conv2"] = lambda: (start(), fired.append(flat._early))[0]
        bs = 6
        g = torch.Generator().manual_seed(17)
        a, b = torch.rand(bs, 1, 112, 112, generator=g), torch.rand(bs, 1, 112, 112, generator=g)
        filenames, partitions, groups = acdc_like_meta(bs)
        tgt = torch.zeros(bs, 1, 1, 1, dtype=torch.long).cuda()
        ep = PretrainEncoderEpocher(model=net, optimizer=torch.optim.SGD([flat.param], lr=0.0), chain_dataloader=iter([]),
                                    num_batches=1, device="cuda", inference_until="Conv5", flat_params=flat)
        ep.add_hooks([hook()])
        net.train()
        with ep.meters.focus_on(ep.meter_focus):
            ep.step_compute(((a.cuda(), b.cuda(), tgt, tgt), filenames, (partitions, groups)), seed=3)
        if overlap:
            assert fired == [True]  # the hook at Conv2's output ran once, during backward, and gathered the early bucket
        grads.append(flat.flat.clone())
        flat.allreduce_()
        assert flat._early is None
    assert torch.equal(grads[0], grads[1])
    assert float(grads[0].abs().max()) > 0


def test_pool_link_ignores_a_gradient_it_did_not_produce():
    """PoolLink: the next block's input-gradient kernel leaves the previous block's BatchNorm-backward sums next to the
    pooled gradient it returns; if the pooled tensor had a second consumer, autograd hands the previous block the SUM of two
    gradients -- a different tensor, or the same one accumulated in place -- and the sums no longer belong to it: the block
    must notice (storage + version counter, the link keeping the tensor alive) and reduce for itself."""
    import spcl_amd  # noqa
    from spcl_amd import functional as F
    from spcl_amd.semi_seg.arch.unet import _ConvBlock
    torch.manual_seed(7)
    a = _ConvBlock(16, 32).cuda().train()
    b = _ConvBlock(32, 64).cuda().train()
    for m in (a, b):
        m._compute_dtype = torch.bfloat16
    x = torch.rand(4, 16, 56, 56, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)

    def run(extra, link):
        for m in (a, b):
            m.zero_grad(set_to_none=True)
        a._plan = (False, True)
        a(x)
        p = a.take_pooled()
        b._plan = (True, False)
        if link:
            b._link_in, a._link_out = a._link_out, None
        else:
            a._link_out = None
        o = b(p)
        loss = o.float().square().mean() + (p.float().square().mean() * 3.0 if extra else 0.0)
        used = []
        real = F._n.call
        F._n.call = lambda name, *args: (used.append((name, args)), real(name, *args))[1]
        try:
            loss.backward()
        finally:
            F._n.call = real
        torch.cuda.synchronize()
        # the link was taken <=> block a did NOT run its own reduction pass over (y, POOLED gradient: the third argument): its
        # sums came with the gradient, as per-tile rows (spcl_bnrelu_pool_backward_rows) or in its accumulator block
        # (spcl_conv3x3_dgrad_poolstats_acc + spcl_bnrelu_backward_acc)
        names = [u[0] for u in used]
        own = (any(nm == "spcl_bnrelu_pool_backward" and args[2] is not None for nm, args in used)
               or any(nm == "spcl_bnrelu_backward_fill_acc" and args[3] is not None for nm, args in used))  # (dpool given)
        taken = not own
        assert (not taken or "spcl_bnrelu_pool_backward_rows" in names or "spcl_bnrelu_backward_rows_acc" in names
                or "spcl_conv3x3_dgrad_poolstats_acc" in names)
        return [q.grad.clone() for q in a.parameters()], taken

    (g1, rows1), (g0, rows0) = run(False, True), run(False, False)
    assert rows1 and not rows0  # the link is taken when the gradient is the producer's own ...
    (h1, hrows1), (h0, hrows0) = run(True, True), run(True, False)
    assert not hrows1 and not hrows0  # ... and not when something else contributed
    for u, v in zip(h1, h0):
        # both reduce for themselves -- the unlinked block through its (clean) accumulator block, the linked one, whose block
        # the consumer's kernel had already added to, through partial rows + the finalize launch: same sums, another order
        assert float((u.float() - v.float()).norm()) <= 2e-3 * float(v.float().norm()) + 1e-12
    for u, v in zip(g1, g0):  # the linked sums are taken in another order: close, not equal
        assert float((u.float() - v.float()).norm()) <= 2e-2 * float(v.float().norm()) + 1e-12
