"""Kernel-level GPU parity through the C ABI (ctypes), one encoder kernel at a time, on identical inputs:
for bf16 the inputs/weights are rounded to bf16 FIRST and the fp32/fp64 CPU reference (torch ops = the oracle's
building blocks) consumes the rounded values, so the only difference left is accumulation order and the final bf16
store."""
from ctypes import c_float

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = {"f32": torch.float32, "bf16": torch.bfloat16}
TOL = {"f32": 2e-5, "bf16": 6e-3}  # relative to max|ref|


def _n():
    import spcl_amd  # noqa
    from spcl_amd import native
    return native


def ru16(c):
    return (c + 15) // 16 * 16


def nhwc(x_nchw, dtype, cs=None):
    N, C, H, W = x_nchw.shape
    cs = cs or ru16(C)
    out = torch.zeros(N, H, W, cs, dtype=dtype, device="cuda")
    out[..., :C] = x_nchw.permute(0, 2, 3, 1).to(dtype)
    return out


def relerr(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rnd(t, dtype):
    return t.to(dtype).float()


def pack(n, w, kind, dtype):
    co, ci = w.shape[:2]
    dtc = n.dtype_code(dtype)
    buf = torch.empty(n.call("spcl_conv_packed_elems", ci, co, kind, dtc), dtype=dtype, device="cuda")
    wc = w.cuda().contiguous()
    n.call("spcl_conv_pack_weights", n.ptr(wc), ci, co, kind, dtc, n.ptr(buf), n.stream())
    return buf


def conv(n, xs, dtype, N, H, W, cin_s, cin_k, cout_s, wp, mode, scale=None, shift=None, stats=False):
    y = torch.empty(N, H, W, cout_s, dtype=dtype, device="cuda")
    st = None
    if stats:
        nt = n.call("spcl_conv_stat_rows", n.dtype_code(dtype), N, H, W, cin_k, cout_s)
        st = torch.empty(n.call("spcl_bn_stats_elems", nt, cout_s), dtype=torch.float32, device="cuda")
        st.ntiles = nt
    n.call("spcl_conv3x3_forward", n.ptr(xs), n.dtype_code(dtype), N, H, W, cin_s, cin_k, cout_s, n.ptr(wp), mode,
           n.ptr(scale), n.ptr(shift), n.ptr(y), n.ptr(st), n.stream())
    return y, st


SHAPES = [(2, 16, 16, 28, 28), (1, 32, 64, 20, 18), (2, 128, 64, 14, 14), (1, 64, 256, 7, 9), (3, 8, 24, 33, 16),
          (1, 256, 256, 14, 14), (2, 32, 32, 21, 42), (1, 64, 128, 35, 14), (1, 16, 32, 126, 28), (1, 32, 16, 238, 14),
          # sizes that are not a multiple of the 14-column tiles: shifted last tiles of the specialised kernels
          (1, 16, 16, 60, 44), (2, 32, 64, 50, 30), (1, 16, 16, 256, 64), (1, 64, 64, 64, 64), (1, 128, 128, 32, 128),
          # the >= 64-channel bf16 layers (workgroup-level GEMM kernel): the encoder's sizes, the 256^2 family's, odd ones
          (2, 64, 64, 56, 56), (2, 64, 128, 28, 28), (3, 128, 128, 28, 28), (2, 128, 256, 14, 14), (2, 256, 256, 16, 16),
          (1, 64, 64, 9, 5), (1, 192, 64, 3, 70), (1, 64, 192, 1, 1), (1, 128, 64, 130, 66)]


GEMM_SHAPES = [s_ for s_ in SHAPES if s_[1] % 64 == 0 and s_[2] % 64 == 0 and max(s_[1], s_[2]) >= 128]


@pytest.fixture
def gemm_conv():
    """the workgroup-level GEMM kernel of the wide bf16 layers forced on at EVERY image size (by default it only takes the
    sizes the per-wave kernels have no specialisation for)"""
    n = _n()
    n.call("spcl_conv_set_gemm", 1)
    yield n
    n.call("spcl_conv_set_gemm", -1)


@pytest.mark.parametrize("N,ci,co,H,W", GEMM_SHAPES)
def test_conv_gemm_forward_dgrad(gemm_conv, N, ci, co, H, W):
    _fwd_body("bf16", N, ci, co, H, W)
    _dgrad_body("bf16", N, ci, co, H, W)


@pytest.mark.parametrize("N,ci,co,H,W", [(2, 128, 128, 28, 28), (1, 256, 64, 14, 14), (1, 64, 128, 30, 17)])
def test_conv_gemm_fused_bnrelu_input(gemm_conv, N, ci, co, H, W):
    _fused_in_body("bf16", N, ci, co, H, W)


@pytest.mark.parametrize("N,C,H,W", [(1, 128, 28, 14), (2, 256, 14, 14), (1, 128, 33, 20)])
def test_conv_gemm_dgrad_with_fused_bn_backward_sums(gemm_conv, N, C, H, W):
    _dgrad_bn_body(N, C, H, W)


@pytest.fixture
def exact_f32():
    """f32 storage on v_mfma_f32_16x16x4_f32 (exact f32 products) instead of the default split-bf16 k-loop
    (spcl_conv_set_f32_split: three bf16 pieces per operand, six bf16 MFMAs per product) -- both hold the f32 tolerance"""
    n = _n()
    assert n.call("spcl_conv_get_f32_split") == 1  # the default
    n.call("spcl_conv_set_f32_split", 0)
    yield n
    n.call("spcl_conv_set_f32_split", 1)


F32_SHAPES = [(2, 16, 16, 28, 28), (1, 32, 64, 20, 18), (2, 128, 64, 14, 14), (3, 8, 24, 33, 16), (1, 256, 256, 14, 14),
              (1, 16, 16, 60, 44), (1, 64, 64, 64, 64), (1, 64, 192, 1, 1)]


@pytest.mark.parametrize("N,ci,co,H,W", F32_SHAPES)
def test_conv_f32_exact_path_forward_dgrad_wgrad(exact_f32, N, ci, co, H, W):
    _fwd_body("f32", N, ci, co, H, W)
    _dgrad_body("f32", N, ci, co, H, W)
    _wgrad_body("f32", N, ci, co, H, W, 0)
    _wgrad_body("f32", N, ci, co, H, W, 1)


def test_conv_f32_exact_path_fused_input_and_image(exact_f32):
    _fused_in_body("f32", 2, 32, 48, 28, 28)
    _fused_in_body("f32", 1, 64, 64, 30, 17)
    for ci in (1, 3):
        _image_mode_body("f32", ci)


@pytest.mark.parametrize("N,ci,co,H,W", [(2, 64, 64, 28, 28), (1, 16, 32, 42, 56)])
def test_conv_f32_split_agrees_with_exact_far_inside_the_tolerance(N, ci, co, H, W):
    """same inputs through both f32 k-loops: the split products drop terms below 2^-24 of a product, the two results differ
    by a few f32 roundings of the accumulated sum -- an order of magnitude inside the 2e-5 both are held to"""
    n = _n()
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, ci, H, W, generator=g)
    w = torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5)
    xs, wp = nhwc(x, torch.float32), pack(n, w, 0, torch.float32)
    ys, _ = conv(n, xs, torch.float32, N, H, W, ci, ci, co, wp, 0)
    n.call("spcl_conv_set_f32_split", 0)
    try:
        ye, _ = conv(n, xs, torch.float32, N, H, W, ci, ci, co, wp, 0)  # (the SAME packed buffer: it carries both layouts)
    finally:
        n.call("spcl_conv_set_f32_split", 1)
    ref = F.conv2d(x.double(), w.double(), None, 1, 1).permute(0, 2, 3, 1).cuda()
    assert relerr(ys, ye) < 2e-6
    assert relerr(ys, ref) < 2e-6 and relerr(ye, ref) < 2e-6


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("N,ci,co,H,W", SHAPES)
def test_conv_forward_raw_and_stats(dt, N, ci, co, H, W):
    _fwd_body(dt, N, ci, co, H, W)


def _fwd_body(dt, N, ci, co, H, W):
    n = _n()
    dtype = DT[dt]
    g = torch.Generator().manual_seed(ci * 1000 + co + H)
    x = rnd(torch.randn(N, ci, H, W, generator=g), dtype)
    w = rnd(torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5), dtype)
    ref = F.conv2d(x.double(), w.double(), None, 1, 1).float()
    cs_i, cs_o = ru16(ci), ru16(co)
    xs, wp = nhwc(x, dtype), pack(n, w, 0, dtype)
    y, st = conv(n, xs, dtype, N, H, W, cs_i, cs_i, cs_o, wp, 0, stats=True)
    got = y[..., :co].permute(0, 3, 1, 2).float().cpu()
    assert relerr(got, ref) < TOL[dt]
    if cs_o > co:
        assert float(y[..., co:].float().abs().max()) == 0.0
    # Chan partials combine to the batch statistics
    st = st[:st.ntiles * 3 * cs_o].view(st.ntiles, 3, cs_o).double().cpu()  # rows [tile][3][CoutS]
    cnt, mean, m2 = st[:, 0, :co], st[:, 1, :co], st[:, 2, :co]
    tot = cnt.sum(0)
    assert int(tot[0]) == N * H * W
    gmean = (cnt * mean).sum(0) / tot
    gm2 = (m2 + cnt * (mean - gmean) ** 2).sum(0)
    np.testing.assert_allclose(gmean.numpy(), ref.double().mean(dim=(0, 2, 3)).numpy(), rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose((gm2 / tot).numpy(), ref.double().var(dim=(0, 2, 3), unbiased=False).numpy(), rtol=2e-3)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("N,ci,co,H,W", [(2, 32, 48, 28, 28), (2, 128, 128, 28, 28), (1, 256, 64, 14, 14),
                                         (1, 64, 64, 30, 17)])
def test_conv_forward_fused_bnrelu_input(dt, N, ci, co, H, W):
    _fused_in_body(dt, N, ci, co, H, W)


def _fused_in_body(dt, N, ci, co, H, W):
    n = _n()
    dtype = DT[dt]
    g = torch.Generator().manual_seed(3)
    x = rnd(torch.randn(N, ci, H, W, generator=g), dtype)
    w = rnd(torch.randn(co, ci, 3, 3, generator=g) / 17, dtype)
    sc, sh = torch.randn(ci, generator=g), torch.randn(ci, generator=g) * 0.3
    act = rnd(torch.relu(x * sc[None, :, None, None] + sh[None, :, None, None]), dtype)  # staged value is rounded
    ref = F.conv2d(act.double(), w.double(), None, 1, 1).float()
    xs, wp, scd, shd = nhwc(x, dtype), pack(n, w, 0, dtype), sc.cuda(), sh.cuda()
    y, _ = conv(n, xs, dtype, N, H, W, ci, ci, co, wp, 1, scd, shd)
    assert relerr(y.permute(0, 3, 1, 2).float().cpu(), ref) < TOL[dt]


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("ci", [1, 2, 3])
def test_conv_forward_image_mode(dt, ci):
    _image_mode_body(dt, ci)


def _image_mode_body(dt, ci):
    n = _n()
    dtype = DT[dt]
    N, co, H, W = 2, 16, 28, 42
    g = torch.Generator().manual_seed(4 + ci)
    x = torch.rand(N, ci, H, W, generator=g)
    w = rnd(torch.randn(co, ci, 3, 3, generator=g) / 3, dtype)
    ref = F.conv2d(rnd(x, dtype).double(), w.double(), None, 1, 1).float()
    xs = x.permute(0, 2, 3, 1).contiguous().cuda()
    wp = pack(n, w, 0, dtype)
    y, _ = conv(n, xs, dtype, N, H, W, ci, 16, 16, wp, 2)
    assert relerr(y.permute(0, 3, 1, 2).float().cpu(), ref) < TOL[dt]


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("N,ci,co,H,W", SHAPES)
def test_conv_dgrad(dt, N, ci, co, H, W):
    _dgrad_body(dt, N, ci, co, H, W)


def _dgrad_body(dt, N, ci, co, H, W):
    n = _n()
    dtype = DT[dt]
    g = torch.Generator().manual_seed(ci + co + W)
    dy = rnd(torch.randn(N, co, H, W, generator=g), dtype)
    w = rnd(torch.randn(co, ci, 3, 3, generator=g) / (3 * co ** 0.5), dtype)
    ref = F.conv_transpose2d(dy.double(), w.double(), None, 1, 1).float()  # == autograd's grad_input
    cs_i, cs_o = ru16(ci), ru16(co)
    dys, wp = nhwc(dy, dtype), pack(n, w, 1, dtype)
    dx, _ = conv(n, dys, dtype, N, H, W, cs_o, cs_o, cs_i, wp, 0)
    assert relerr(dx[..., :ci].permute(0, 3, 1, 2).float().cpu(), ref) < TOL[dt]


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("N,ci,co,H,W", SHAPES + [(4, 16, 16, 56, 56)])
@pytest.mark.parametrize("mode", [0, 1])
def test_conv_wgrad(dt, N, ci, co, H, W, mode):
    _wgrad_body(dt, N, ci, co, H, W, mode)


def _wgrad_body(dt, N, ci, co, H, W, mode):
    n = _n()
    dtype = DT[dt]
    g = torch.Generator().manual_seed(ci * 7 + co + H)
    x = rnd(torch.randn(N, ci, H, W, generator=g), dtype)
    dy = rnd(torch.randn(N, co, H, W, generator=g), dtype)
    sc, sh = torch.randn(ci, generator=g), torch.randn(ci, generator=g) * 0.3
    xin = rnd(torch.relu(x * sc[None, :, None, None] + sh[None, :, None, None]), dtype) if mode == 1 else x
    ref = torch.nn.grad.conv2d_weight(xin.double(), (co, ci, 3, 3), dy.double(), 1, 1).float()
    cs_i, cs_o = ru16(ci), ru16(co)
    scp = torch.zeros(cs_i)
    shp = torch.zeros(cs_i)
    scp[:ci], shp[:ci] = sc, sh
    ws = torch.empty(n.call("spcl_conv_wgrad_workspace_bytes", N, H, W, cs_i, cs_o) // 4, device="cuda")
    dw = torch.empty(co, ci, 3, 3, device="cuda")
    xs, dys, scd, shd = nhwc(x, dtype), nhwc(dy, dtype), scp.cuda(), shp.cuda()  # keep alive: ptr() borrows
    n.call("spcl_conv3x3_wgrad", n.ptr(xs), n.ptr(dys), n.dtype_code(dtype), N, H, W, ci, cs_i,
           cs_i, co, cs_o, mode, n.ptr(scd), n.ptr(shd), n.ptr(ws), n.ptr(dw), n.stream())
    assert relerr(dw.cpu(), ref) < (2e-5 if dt == "f32" else 2e-3)


def _wgrad_item(n, N, ci, co, H, W, mode, seed, keep):
    """one layer of a batched weight-gradient launch + its fp64 reference on the same bf16-rounded operands"""
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(seed)
    x = rnd(torch.randn(N, ci, H, W, generator=g), dtype)
    dy = rnd(torch.randn(N, co, H, W, generator=g), dtype)
    sc, sh = torch.randn(ci, generator=g), torch.randn(ci, generator=g) * 0.3
    xin = rnd(torch.relu(x * sc[None, :, None, None] + sh[None, :, None, None]), dtype) if mode == 1 else x
    ref = torch.nn.grad.conv2d_weight(xin.double(), (co, ci, 3, 3), dy.double(), 1, 1).float()
    xs, dys, scd, shd = nhwc(x, dtype), nhwc(dy, dtype), sc.cuda(), sh.cuda()
    dw = torch.full((co, ci, 3, 3), 0.25, device="cuda")
    keep += [xs, dys, scd, shd, dw]
    it = n.WgradItem(xs.data_ptr(), dys.data_ptr(), scd.data_ptr() if mode else None, shd.data_ptr() if mode else None,
                     dw.data_ptr(), N, H, W, ci, ci, co, co, mode)
    return it, dw, ref


@pytest.mark.parametrize("accumulate", [0, 1])
@pytest.mark.parametrize("layers", [
    # the five >=64-channel encoder layers of a 224x224 step at reduced batch (56 / 28 / 14 squares, both input modes)
    [(4, 64, 64, 56, 56, 1), (4, 64, 128, 28, 28, 0), (4, 128, 128, 28, 28, 1), (4, 128, 256, 14, 14, 0),
     (4, 256, 256, 14, 14, 1)],
    # 256x256 inputs (64 / 32 / 16 squares: 16-row tiles), a ragged one, a single tiny one
    [(2, 64, 64, 64, 64, 1), (2, 64, 128, 32, 32, 0), (2, 256, 256, 16, 16, 1)],
    [(3, 64, 192, 23, 37, 1), (1, 128, 64, 5, 70, 0)],
    [(1, 64, 64, 3, 3, 1)],
    # eight layers (the batch limit), decoder-like widths
    [(1, 64 * (1 + i % 3), 64 * (1 + (i + 1) % 2), 14 + 7 * (i % 3), 14 + i, i % 2) for i in range(8)],
])
def test_conv_wgrad_batched(layers, accumulate):
    """several layers' weight gradients in ONE launch (csrc/wgrad_gemm.hip) vs fp64 on identical bf16 operands; with
    accumulate the result is added to what the gradient buffer held (0.25 here)."""
    n = _n()
    keep, items, outs = [], [], []
    for k, (N, ci, co, H, W, mode) in enumerate(layers):
        it, dw, ref = _wgrad_item(n, N, ci, co, H, W, mode, 100 + k, keep)
        items.append(it)
        outs.append((dw, ref))
    arr = (n.WgradItem * len(items))(*items)
    nbytes = n.call("spcl_conv_wgrad_batched_workspace_bytes", arr, len(items))
    assert nbytes > 0
    ws = torch.empty(nbytes // 4, device="cuda")
    n.call("spcl_conv3x3_wgrad_batched", arr, len(items), accumulate, n.ptr(ws), n.stream())
    for k, (dw, ref) in enumerate(outs):
        got = dw.cpu() - (0.25 if accumulate else 0.0)
        assert relerr(got, ref) < 2e-3, (k, layers[k], relerr(got, ref))
    # bit-for-bit repeatable (fixed-order reduction, no atomics)
    first = [dw.clone() for dw, _ in outs]
    for dw, _ in outs:
        dw.fill_(0.25)
    n.call("spcl_conv3x3_wgrad_batched", arr, len(items), accumulate, n.ptr(ws), n.stream())
    for a, (dw, _) in zip(first, outs):
        assert torch.equal(a, dw)


@pytest.mark.parametrize("with_wide", [False, True])
@pytest.mark.parametrize("accumulate", [0, 1])
def test_conv_wgrad_tails_ride_in_the_batched_reduction(with_wide, accumulate):
    """The narrow layers' final sums (and the first layer's rows of the fused BN-backward + weight-gradient pass) captured
    as tails and finished by the batched launch's reduction kernel equal what their own reduction launches give (only
    the summation tree differs), with and without >= 64-channel items in the same launch."""
    import ctypes
    n = _n()
    keep, tails, expect = [], [], []
    g = torch.Generator().manual_seed(77)
    for (dt, N, ci, co, H, W, mode) in [("bf16", 3, 16, 16, 56, 56, 1), ("bf16", 2, 16, 32, 28, 42, 0),
                                        ("bf16", 2, 32, 32, 28, 28, 1), ("bf16", 2, 32, 64, 14, 14, 0),
                                        ("f32", 2, 24, 40, 14, 21, 0), ("bf16", 2, 1, 16, 30, 44, 2)]:
        dtype = DT[dt]
        x = rnd(torch.randn(N, ci, H, W, generator=g), dtype) if mode != 2 else torch.rand(N, 1, H, W, generator=g)
        dy = rnd(torch.randn(N, co, H, W, generator=g), dtype)
        cs_i, cs_o = ru16(ci), ru16(co)
        sc, sh = torch.zeros(cs_i), torch.zeros(cs_i)
        sc[:ci], sh[:ci] = torch.randn(ci, generator=g), torch.randn(ci, generator=g) * 0.3
        xs = nhwc(x, dtype) if mode != 2 else x.permute(0, 2, 3, 1).contiguous().cuda()
        dys, scd, shd = nhwc(dy, dtype), sc.cuda(), sh.cuda()
        nb = n.call("spcl_conv_wgrad_workspace_bytes", N, H, W, cs_i, cs_o)
        ws, ws2 = torch.empty(nb // 4, device="cuda"), torch.empty(nb // 4, device="cuda")
        dw_own = torch.empty(co, ci, 3, 3, device="cuda")
        dw_tail = torch.full((co, ci, 3, 3), 0.25, device="cuda")
        args = lambda w_, d_: (n.ptr(xs), n.ptr(dys), n.dtype_code(dtype), N, H, W, ci, ci if mode == 2 else cs_i, cs_i,
                               co, cs_o, mode, n.ptr(scd) if mode == 1 else None, n.ptr(shd) if mode == 1 else None,
                               n.ptr(w_), n.ptr(d_), n.stream())
        n.call("spcl_conv3x3_wgrad", *args(ws, dw_own))
        t = n.WgradTail()
        n.call("spcl_wgrad_tail_capture", ctypes.byref(t))
        n.call("spcl_conv3x3_wgrad", *args(ws2, dw_tail))
        assert t.kind == 0 and t.dw == dw_tail.data_ptr()
        assert torch.all(dw_tail == 0.25)  # untouched until the batched launch
        keep += [xs, dys, scd, shd, ws, ws2]
        tails.append(t)
        expect.append((dw_own, dw_tail))
    # a capture is one-shot: the next producer runs its own reduction again
    n.call("spcl_conv3x3_wgrad", *args(ws, dw_own))
    # kind 1: the first layer's rows
    N, C, H, W = 2, 16, 30, 44
    dtype, dtc, cs = torch.bfloat16, n.dtype_code(torch.bfloat16), 16
    x = torch.rand(N, 1, H, W, generator=g)
    y = rnd(F.conv2d(x, torch.randn(C, 1, 3, 3, generator=g) * 0.5, padding=1), dtype)
    dact = rnd(torch.randn(N, C, H, W, generator=g), dtype)
    st = torch.zeros(4, cs)
    st[0, :C], st[1, :C] = y.mean(dim=(0, 2, 3)), 1.0 / torch.sqrt(y.var(dim=(0, 2, 3), unbiased=False) + 1e-5)
    st[2, :C], st[3, :C] = st[1, :C], -st[0, :C] * st[1, :C]
    st = st.cuda()
    ys, das, xs = nhwc(y, dtype), nhwc(dact, dtype), x.permute(0, 2, 3, 1).contiguous().cuda()
    nb = n.call("spcl_bnrelu_image_wgrad_workspace_bytes", N, H, W, cs)
    wsa, wsb = torch.empty(nb // 4, device="cuda"), torch.empty(nb // 4, device="cuda")
    dgm, dbt = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    dw_own, dw_tail = torch.empty(C, 1, 3, 3, device="cuda"), torch.full((C, 1, 3, 3), 0.25, device="cuda")
    iargs = lambda w_, d_: (n.ptr(ys), n.ptr(das), n.ptr(xs), dtc, N, H, W, C, cs, n.ptr(st[0]), n.ptr(st[1]),
                            n.ptr(st[2]), n.ptr(st[3]), 1, n.ptr(w_), n.ptr(dgm), n.ptr(dbt), n.ptr(d_), n.stream())
    n.call("spcl_bnrelu_backward_image_wgrad", *iargs(wsa, dw_own))
    t = n.WgradTail()
    n.call("spcl_wgrad_tail_capture", ctypes.byref(t))
    n.call("spcl_bnrelu_backward_image_wgrad", *iargs(wsb, dw_tail))
    assert t.kind == 1
    tails.append(t)
    expect.append((dw_own, dw_tail))
    items, outs = [], []
    if with_wide:
        for k, (N_, ci, co, H_, W_, mode) in enumerate([(2, 64, 64, 28, 28, 1), (2, 128, 64, 14, 14, 0)]):
            it, dw, ref = _wgrad_item(n, N_, ci, co, H_, W_, mode, 300 + k, keep)
            items.append(it)
            outs.append((dw, ref))
    arr = (n.WgradItem * len(items))(*items) if items else None
    wsw = torch.empty(n.call("spcl_conv_wgrad_batched_workspace_bytes", arr, len(items)) // 4, device="cuda") \
        if items else None
    tarr = (n.WgradTail * len(tails))(*tails)
    n.call("spcl_conv3x3_wgrad_batched_tails", arr, len(items), tarr, len(tails), accumulate, n.ptr(wsw), n.stream())
    for k, (own, tl) in enumerate(expect):
        got = tl - (0.25 if accumulate else 0.0)
        assert relerr(got.cpu(), own.cpu()) < 2e-6, (k, relerr(got.cpu(), own.cpu()))
    for k, (dw, ref) in enumerate(outs):
        assert relerr(dw.cpu() - (0.25 if accumulate else 0.0), ref) < 2e-3
    with pytest.raises(RuntimeError):  # an empty / foreign descriptor is refused
        bad = (n.WgradTail * 1)(n.WgradTail())
        n.call("spcl_conv3x3_wgrad_batched_tails", None, 0, bad, 1, 0, None, n.stream())


def test_conv_wgrad_batched_rejects_what_it_cannot_do():
    n = _n()
    keep = []
    it, _, _ = _wgrad_item(n, 1, 64, 64, 8, 8, 0, 1, keep)
    bad = n.WgradItem(it.x, it.dy, None, None, it.dw_oihw, 1, 8, 8, 48, 48, 64, 64, 0)  # 48 input channels
    arr = (n.WgradItem * 1)(bad)
    assert n.call("spcl_conv_wgrad_batched_workspace_bytes", arr, 1) == 0
    assert not n.call("spcl_conv_wgrad_batched_supported", n.SPCL_F32, 64, 64, 64, 64, 0)
    assert not n.call("spcl_conv_wgrad_batched_supported", n.SPCL_BF16, 64, 64, 64, 64, 2)
    ws = torch.empty(1024, device="cuda")
    with pytest.raises(RuntimeError):
        n.call("spcl_conv3x3_wgrad_batched", arr, 1, 0, n.ptr(ws), n.stream())


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_conv_wgrad_image_mode(dt):
    n = _n()
    dtype = DT[dt]
    N, ci, co, H, W = 3, 1, 16, 30, 44
    g = torch.Generator().manual_seed(9)
    x = torch.rand(N, ci, H, W, generator=g)
    dy = rnd(torch.randn(N, co, H, W, generator=g), dtype)
    ref = torch.nn.grad.conv2d_weight(rnd(x, dtype).double(), (co, ci, 3, 3), dy.double(), 1, 1).float()
    ws = torch.empty(n.call("spcl_conv_wgrad_workspace_bytes", N, H, W, 16, 16) // 4, device="cuda")
    dw = torch.empty(co, ci, 3, 3, device="cuda")
    xs, dys = x.permute(0, 2, 3, 1).contiguous().cuda(), nhwc(dy, dtype)
    n.call("spcl_conv3x3_wgrad", n.ptr(xs), n.ptr(dys),
           n.dtype_code(dtype), N, H, W, ci, ci, 16, co, 16, 2, None, None, n.ptr(ws), n.ptr(dw), n.stream())
    assert relerr(dw.cpu(), ref) < (2e-5 if dt == "f32" else 2e-3)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("N,C,H,W,pool,with_act", [(2, 16, 28, 28, True, False), (2, 48, 14, 10, True, True),
                                                   (3, 32, 7, 5, True, True), (2, 64, 14, 14, False, True),
                                                   (1, 24, 9, 9, True, False)])
def test_bn_relu_pool_forward_backward(dt, N, C, H, W, pool, with_act):
    """BatchNorm(train) -> ReLU -> (MaxPool 2x2, floor) forward and backward against torch autograd on the same y."""
    n = _n()
    dtype = DT[dt]
    dtc = n.dtype_code(dtype)
    cs = ru16(C)
    g = torch.Generator().manual_seed(C + H)
    y = rnd(torch.randn(N, C, H, W, generator=g) * 1.5 + 0.3, dtype)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    # reference
    yr = y.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    a = F.relu(F.batch_norm(yr, None, None, gr, br, True, 0.1, 1e-5))
    outs = []
    if with_act or not pool:
        outs.append(a)
    if pool:
        outs.append(F.max_pool2d(a, 2, 2))
    douts = [rnd(torch.randn(o.shape, generator=g), dtype).double() for o in outs]
    torch.autograd.backward(outs, douts)
    # HIP statistics through bn_finalize from a single exact partial
    mean = y.double().mean(dim=(0, 2, 3))
    var = y.double().var(dim=(0, 2, 3), unbiased=False)
    stats = torch.zeros(1, 3, cs)
    stats[0, 0, :C] = N * H * W
    stats[0, 1, :C] = mean.float()
    stats[0, 2, :C] = (var * N * H * W).float()
    rm, rv = torch.zeros(C).cuda(), torch.ones(C).cuda()
    nbt = torch.zeros((), dtype=torch.long).cuda()
    st = torch.empty(4, cs, device="cuda")
    stats_d, gamma_d, beta_d = stats.cuda(), gamma.cuda(), beta.cuda()
    n.call("spcl_bn_finalize", n.ptr(stats_d), 1, C, cs, n.ptr(gamma_d), n.ptr(beta_d), c_float(0.1),
           c_float(1e-5), n.ptr(rm), n.ptr(rv), n.ptr(nbt), n.ptr(st[0]), n.ptr(st[1]), n.ptr(st[2]), n.ptr(st[3]),
           n.stream())
    assert int(nbt) == 1
    np.testing.assert_allclose(rm.cpu().numpy(), 0.1 * mean.float().numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(rv.cpu().numpy(),
                               (0.9 + 0.1 * y.double().var(dim=(0, 2, 3), unbiased=True)).float().numpy(), rtol=1e-5)
    ys = nhwc(y, dtype)
    act = torch.empty(N, H, W, cs, dtype=dtype, device="cuda") if (with_act or not pool) else None
    pl = torch.empty(N, H // 2, W // 2, cs, dtype=dtype, device="cuda") if pool else None
    n.call("spcl_bnrelu_pool_forward", n.ptr(ys), dtc, N, H, W, cs, n.ptr(st[2]), n.ptr(st[3]), n.ptr(act), n.ptr(pl),
           n.stream())
    k = 0
    if act is not None:
        assert relerr(act[..., :C].permute(0, 3, 1, 2).float().cpu(), outs[k].detach().float()) < TOL[dt]
        k += 1
    if pl is not None:
        assert relerr(pl[..., :C].permute(0, 3, 1, 2).float().cpu(), outs[k].detach().float()) < TOL[dt]
    # backward
    k = 0
    dact = dpool = None
    if act is not None:
        dact = nhwc(douts[k].float(), dtype)
        k += 1
    if pl is not None:
        dpool = nhwc(douts[k].float(), dtype)
    ws = torch.empty(n.call("spcl_bnrelu_bwd_workspace_bytes", N, H, W, cs) // 4, device="cuda")
    dgm, dbt = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    dy = torch.empty(N, H, W, cs, dtype=dtype, device="cuda")
    n.call("spcl_bnrelu_pool_backward", n.ptr(ys), n.ptr(dact), n.ptr(dpool), dtc, N, H, W, C, cs, n.ptr(st[0]),
           n.ptr(st[1]), n.ptr(st[2]), n.ptr(st[3]), 1, n.ptr(ws), n.ptr(dgm), n.ptr(dbt), n.ptr(dy), n.stream())
    tol = 5e-5 if dt == "f32" else 8e-3
    assert relerr(dgm.cpu(), gr.grad.float()) < tol
    assert relerr(dbt.cpu(), br.grad.float()) < tol
    assert relerr(dy[..., :C].permute(0, 3, 1, 2).float().cpu(), yr.grad.float()) < tol


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("N,C,H,W", [(2, 16, 30, 44), (3, 8, 17, 9), (1, 32, 5, 300), (2, 16, 224, 224)])
def test_bn_relu_backward_fused_image_wgrad(dt, N, C, H, W):
    """conv3x3(1 -> C) -> BatchNorm(train) -> ReLU backward with dy consumed in registers by dW: against torch autograd
    (float64) on the same raw conv output y, and against the two-launch HIP path it replaces."""
    n = _n()
    dtype = DT[dt]
    dtc = n.dtype_code(dtype)
    cs = ru16(C)
    g = torch.Generator().manual_seed(C + H)
    x = torch.rand(N, 1, H, W, generator=g)
    w = torch.randn(C, 1, 3, 3, generator=g) * 0.5
    y = rnd(F.conv2d(x, w, padding=1), dtype)  # the raw conv output as the forward pass stored it
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    dact = rnd(torch.randn(N, C, H, W, generator=g), dtype)
    # reference: dy from BN+ReLU backward at y, dW = correlation of dy with the image
    yr = y.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    F.relu(F.batch_norm(yr, None, None, gr, br, True, 0.1, 1e-5)).backward(dact.double())
    dw_ref = torch.nn.grad.conv2d_weight(x.double(), (C, 1, 3, 3), yr.grad, 1, 1).float()
    mean = y.double().mean(dim=(0, 2, 3))
    var = y.double().var(dim=(0, 2, 3), unbiased=False)
    st = torch.zeros(4, cs)
    st[0, :C], st[1, :C] = mean.float(), (1.0 / torch.sqrt(var + 1e-5)).float()
    st[2, :C] = gamma * st[1, :C]
    st[3, :C] = beta - st[0, :C] * st[2, :C]
    st = st.cuda()
    ys, das, xs = nhwc(y, dtype), nhwc(dact, dtype), x.permute(0, 2, 3, 1).contiguous().cuda()
    ws = torch.empty(n.call("spcl_bnrelu_image_wgrad_workspace_bytes", N, H, W, cs) // 4, device="cuda")
    dgm, dbt = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    dw = torch.full((C, 1, 3, 3), float("nan"), device="cuda")
    n.call("spcl_bnrelu_backward_image_wgrad", n.ptr(ys), n.ptr(das), n.ptr(xs), dtc, N, H, W, C, cs, n.ptr(st[0]),
           n.ptr(st[1]), n.ptr(st[2]), n.ptr(st[3]), 1, n.ptr(ws), n.ptr(dgm), n.ptr(dbt), n.ptr(dw), n.stream())
    tol = 5e-5 if dt == "f32" else 8e-3
    assert relerr(dgm.cpu(), gr.grad.float()) < tol
    assert relerr(dbt.cpu(), br.grad.float()) < tol
    assert relerr(dw.cpu(), dw_ref) < (1e-4 if dt == "f32" else 8e-3)
    # the unfused pair on the same inputs (its dy is rounded to dtype before the contraction; the fused one is not)
    ws2 = torch.empty(n.call("spcl_bnrelu_bwd_workspace_bytes", N, H, W, cs) // 4, device="cuda")
    dg2, db2 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    dy = torch.empty(N, H, W, cs, dtype=dtype, device="cuda")
    n.call("spcl_bnrelu_pool_backward", n.ptr(ys), n.ptr(das), None, dtc, N, H, W, C, cs, n.ptr(st[0]), n.ptr(st[1]),
           n.ptr(st[2]), n.ptr(st[3]), 1, n.ptr(ws2), n.ptr(dg2), n.ptr(db2), n.ptr(dy), n.stream())
    ws3 = torch.empty(n.call("spcl_conv_wgrad_workspace_bytes", N, H, W, 16, cs) // 4, device="cuda")
    dw2 = torch.empty(C, 1, 3, 3, device="cuda")
    n.call("spcl_conv3x3_wgrad", n.ptr(xs), n.ptr(dy), dtc, N, H, W, 1, 1, 16, C, cs, 2, None, None, n.ptr(ws3),
           n.ptr(dw2), n.stream())
    assert torch.equal(dgm, dg2) and torch.equal(dbt, db2)  # same reduction kernels
    assert relerr(dw2.cpu(), dw_ref) < (1e-4 if dt == "f32" else 1e-1)
    # deterministic
    dw3 = torch.empty_like(dw)
    n.call("spcl_bnrelu_backward_image_wgrad", n.ptr(ys), n.ptr(das), n.ptr(xs), dtc, N, H, W, C, cs, n.ptr(st[0]),
           n.ptr(st[1]), n.ptr(st[2]), n.ptr(st[3]), 1, n.ptr(ws), n.ptr(dgm), n.ptr(dbt), n.ptr(dw3), n.stream())
    assert torch.equal(dw, dw3)


@pytest.mark.parametrize("ntiles,C", [(3000, 64), (1500, 32), (700, 64), (5000, 48), (2048, 128), (16384, 16)])
def test_bn_finalize_many_tiles_all_launch_shapes(ntiles, C):
    """spcl_bn_finalize from many per-tile (count, mean, M2) rows: the one-launch (16- and 4-channel workgroups) and the
    two-level paths against float64 pooling of the same rows."""
    n = _n()
    cs = ru16(C)
    g = torch.Generator().manual_seed(ntiles + C)
    cnt = torch.randint(90, 197, (ntiles, 1), generator=g).double().expand(ntiles, cs).clone()
    mean = torch.randn(ntiles, cs, generator=g).double() * 0.5 + 0.2
    m2 = (torch.rand(ntiles, cs, generator=g).double() + 0.1) * cnt
    rows = torch.stack([cnt, mean, m2], dim=1).float()  # [ntiles][3][cs]
    stats = torch.zeros(n.call("spcl_bn_stats_elems", ntiles, cs))
    stats[:rows.numel()] = rows.flatten()
    r64 = rows.double()
    N = r64[:, 0].sum(0)
    mu = (r64[:, 0] * r64[:, 1]).sum(0) / N
    var = (r64[:, 2] + r64[:, 0] * r64[:, 1] ** 2).sum(0) / N - mu ** 2
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    st = torch.empty(4, cs, device="cuda")
    stats_d, gamma_d, beta_d = stats.cuda(), gamma.cuda(), beta.cuda()
    n.call("spcl_bn_finalize", n.ptr(stats_d), ntiles, C, cs, n.ptr(gamma_d), n.ptr(beta_d), c_float(0.1),
           c_float(1e-5), None, None, None, n.ptr(st[0]), n.ptr(st[1]), n.ptr(st[2]), n.ptr(st[3]), n.stream())
    np.testing.assert_allclose(st[0, :C].cpu().numpy(), mu[:C].float().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(st[1, :C].cpu().numpy(), (1 / torch.sqrt(var[:C] + 1e-5)).float().numpy(), rtol=1e-5)
    assert float(st[:, C:].abs().max()) == 0.0 if cs > C else True
    # the two-level path is ONE launch whose last group finishes (self-resetting tickets): again and again the same bits
    for _ in range(3):
        st2 = torch.empty(4, cs, device="cuda")
        stats_d = stats.cuda()
        n.call("spcl_bn_finalize", n.ptr(stats_d), ntiles, C, cs, n.ptr(gamma_d), n.ptr(beta_d), c_float(0.1),
               c_float(1e-5), None, None, None, n.ptr(st2[0]), n.ptr(st2[1]), n.ptr(st2[2]), n.ptr(st2[3]), n.stream())
        assert torch.equal(st, st2)


@pytest.mark.parametrize("N,C,H,W", [(2, 16, 28, 28), (1, 32, 56, 28), (2, 64, 14, 14), (1, 128, 28, 14), (3, 16, 224, 42),
                                     (9, 16, 224, 224), (2, 256, 14, 14), (1, 64, 56, 56), (1, 128, 33, 20)])
def test_dgrad_with_fused_bn_backward_sums(N, C, H, W):
    _dgrad_bn_body(N, C, H, W)


def _dgrad_bn_body(N, C, H, W):
    """spcl_conv3x3_dgrad_bnstats + spcl_bnrelu_backward_rows against the three-kernel path they replace (plain dgrad,
    BN-backward reduction pass, apply): g bit-identical, dgamma / dbeta / dy equal up to summation order."""
    n = _n()
    dtype, dtc, cs = torch.bfloat16, 1, ru16(C)
    if not n.call("spcl_conv_dgrad_bnstats_supported", dtc, N, H, W, cs, cs):
        pytest.skip("no specialised kernel for this shape")
    g_ = torch.Generator().manual_seed(C + H)
    dy_in = nhwc(rnd(torch.randn(N, C, H, W, generator=g_), dtype), dtype)
    y2 = nhwc(rnd(torch.randn(N, C, H, W, generator=g_) * 1.3 + 0.2, dtype), dtype)
    w = torch.randn(C, C, 3, 3, generator=g_) / (3.0 * C ** 0.5)
    wp_t = pack(n, w, 1, dtype)
    st = torch.zeros(4, cs)
    st[0, :C] = torch.randn(C, generator=g_) * 0.1 + 0.2          # mean
    st[1, :C] = torch.rand(C, generator=g_) + 0.5                 # invstd
    st[2, :C] = st[1, :C] * (torch.rand(C, generator=g_) + 0.5)   # scale = gamma * invstd
    st[3, :C] = torch.randn(C, generator=g_) * 0.2 - st[0, :C] * st[2, :C]
    st = st.cuda()
    # reference path: plain dgrad, then the BN backward with its own reduction pass
    g_ref, _ = conv(n, dy_in, dtype, N, H, W, cs, cs, cs, wp_t, 0)
    ws = torch.empty(n.call("spcl_bnrelu_bwd_workspace_bytes", N, H, W, cs) // 4, device="cuda")
    dg0, db0 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    dy0 = torch.empty(N, H, W, cs, dtype=dtype, device="cuda")
    n.call("spcl_bnrelu_pool_backward", n.ptr(y2), n.ptr(g_ref), None, dtc, N, H, W, C, cs, n.ptr(st[0]), n.ptr(st[1]),
           n.ptr(st[2]), n.ptr(st[3]), 1, n.ptr(ws), n.ptr(dg0), n.ptr(db0), n.ptr(dy0), n.stream())
    # fused path
    nt = n.call("spcl_conv_stat_rows", dtc, N, H, W, cs, cs)
    g1 = torch.empty(N, H, W, cs, dtype=dtype, device="cuda")
    rows = torch.full((nt * 2 * cs,), float("nan"), device="cuda")
    n.call("spcl_conv3x3_dgrad_bnstats", n.ptr(dy_in), dtc, N, H, W, cs, cs, n.ptr(wp_t), n.ptr(g1), n.ptr(y2),
           n.ptr(st[2]), n.ptr(st[3]), n.ptr(st[0]), n.ptr(rows), n.stream())
    assert torch.equal(g1, g_ref)
    assert not torch.isnan(rows).any()
    dg1, db1 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    dy1 = torch.empty(N, H, W, cs, dtype=dtype, device="cuda")
    n.call("spcl_bnrelu_backward_rows", n.ptr(y2), n.ptr(g1), None, n.ptr(rows), nt, dtc, N, H, W, C, cs, n.ptr(st[0]),
           n.ptr(st[1]), n.ptr(st[2]), n.ptr(st[3]), 1, n.ptr(ws), n.ptr(dg1), n.ptr(db1), n.ptr(dy1), None, n.stream())
    assert relerr(dg1, dg0) < 2e-5 and relerr(db1, db0) < 2e-5
    assert relerr(dy1.float(), dy0.float()) < 8e-3
    # deterministic
    rows2 = torch.empty_like(rows)
    n.call("spcl_conv3x3_dgrad_bnstats", n.ptr(dy_in), dtc, N, H, W, cs, cs, n.ptr(wp_t), n.ptr(g1), n.ptr(y2),
           n.ptr(st[2]), n.ptr(st[3]), n.ptr(st[0]), n.ptr(rows2), n.stream())
    assert torch.equal(rows, rows2)


def test_fused_radam_matches_torch_radam():
    """spcl_radam_step == torch.optim.RAdam (CPU, single tensor) over the un-rectified (rho_t <= 5) and rectified
    steps, with weight decay and a learning-rate change in between."""
    from spcl_amd.optim import FusedRAdam
    g = torch.Generator().manual_seed(5)
    n = 10007  # not a multiple of 4: exercises the tail
    p0 = torch.randn(n, generator=g)
    ref_p = torch.nn.Parameter(p0.clone())
    hip_p = torch.nn.Parameter(p0.clone().cuda())
    ref = torch.optim.RAdam([ref_p], lr=2e-3, weight_decay=1e-2, foreach=False)
    hip = FusedRAdam([hip_p], lr=2e-3, weight_decay=1e-2)
    for it in range(9):
        grad = torch.randn(n, generator=g) * (1.0 + it)
        if it == 6:
            for o in (ref, hip):
                o.param_groups[0]["lr"] = 5e-4
        ref_p.grad = grad.clone()
        hip_p.grad = grad.clone().cuda()
        ref.step()
        hip.step()
        np.testing.assert_allclose(hip_p.detach().cpu().numpy(), ref_p.detach().numpy(), rtol=2e-6, atol=2e-7)
    st = hip.state[hip_p]
    assert int(st["step"]) == 9
    np.testing.assert_allclose(st["exp_avg_sq"].cpu().numpy(), ref.state[ref_p]["exp_avg_sq"].numpy(), rtol=1e-5)
    sd = hip.state_dict()
    hip2 = FusedRAdam([hip_p], lr=2e-3, weight_decay=1e-2)
    hip2.load_state_dict(sd)
    assert hip2.state[hip_p]["step"].dtype == torch.int64 and int(hip2.state[hip_p]["step"]) == 9


@pytest.mark.parametrize("shape,dtype", [((32, 1, 224, 224), torch.float32), ((6, 3, 14, 10), torch.bfloat16),
                                         ((5, 2, 7, 9), torch.float32)])
def test_flip_batch_matches_per_sample_flips(shape, dtype):
    """spcl_flip_batch == stack([flip(sample) ...]) with the reference's per-sample random stream (same seed)."""
    from spcl_amd.semi_seg.epochers.helper import FixRandomSeed, TensorRandomFlip
    f = TensorRandomFlip(axis=[1, 2], threshold=0.8)
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(1)).to(dtype)
    with FixRandomSeed(11):
        ref = torch.stack([f(s) for s in x], dim=0)
    with FixRandomSeed(11):
        got = f.apply_batch(x.cuda())
    assert torch.equal(got.cpu(), ref)
    out = torch.empty(2 * shape[0], *shape[1:], dtype=dtype, device="cuda")
    with FixRandomSeed(11):
        f.apply_batch(x.cuda(), out=out[shape[0]:])
    assert torch.equal(out[shape[0]:].cpu(), ref)


def test_builtin_kernel_timer_reports_symbols_and_costs():
    """spcl_profile_*: one record per kernel launch with the rocprofv3-style symbol, a positive duration and the
    algorithmic bytes / FLOPs declared by the entry point."""
    import ctypes
    n = _n()
    dtype = torch.bfloat16
    N, ci, co, H, W = 2, 16, 16, 28, 28
    x = torch.randn(N, ci, H, W)
    w = torch.randn(co, ci, 3, 3) / 12
    xs, wp = nhwc(x, dtype), pack(n, w, 0, dtype)
    n.call("spcl_profile_enable", 1)
    try:
        conv(n, xs, dtype, N, H, W, 16, 16, 16, wp, 0, stats=True)
        torch.cuda.synchronize()
        assert n.call("spcl_profile_count") == 1
        name = ctypes.create_string_buffer(256)
        us, by, fl = ctypes.c_float(), ctypes.c_double(), ctypes.c_double()
        n.call("spcl_profile_get", 0, name, 256, ctypes.byref(us), ctypes.byref(by), ctypes.byref(fl))
        assert name.value.decode().startswith("spcl::conv3x3_") and "(" not in name.value.decode()
        assert us.value > 0
        assert by.value == N * H * W * (16 + 16) * 2 + 9 * 16 * 16 * 2
        assert fl.value == 2.0 * N * H * W * 9 * 16 * 16
    finally:
        n.call("spcl_profile_enable", 0)
    assert n.call("spcl_profile_count") == 0


@pytest.mark.parametrize("neg_scales", [False, True])
@pytest.mark.parametrize("N,ci,co,H2,W2", [(2, 16, 32, 56, 56), (1, 32, 64, 28, 56), (2, 64, 128, 28, 28), (1, 128, 256, 28, 28),
                                           (3, 16, 32, 224, 42), (1, 16, 32, 57, 31)])
def test_dgrad_with_fused_pooled_bn_backward_sums(N, ci, co, H2, W2, neg_scales):
    """spcl_conv3x3_dgrad_poolstats + spcl_bnrelu_pool_backward_rows against the path they replace (plain dgrad of the next
    block's first conv, then spcl_bnrelu_pool_backward with its own reduction pass over y2): the input gradient g is
    bit-identical, dgamma / dbeta / dy equal up to summation order.  ci = channels of the pooled layer, co = of the conv."""
    n = _n()
    dtype, dtc = torch.bfloat16, 1
    H, W = H2 // 2, W2 // 2
    if not n.call("spcl_conv_dgrad_poolstats_supported", dtc, N, H, W, co, ci, H2, W2):
        pytest.skip("no specialised kernel for this shape")
    g_ = torch.Generator().manual_seed(ci + H2)
    dy_in = nhwc(rnd(torch.randn(N, co, H, W, generator=g_), dtype), dtype)            # grad of the conv's output
    y2 = nhwc(rnd(torch.randn(N, ci, H2, W2, generator=g_) * 1.3 + 0.2, dtype), dtype)  # pooled layer's raw output
    w = torch.randn(co, ci, 3, 3, generator=g_) / (3.0 * co ** 0.5)
    wp_t = pack(n, w, 1, dtype)
    st = torch.zeros(4, ci)
    st[0] = torch.randn(ci, generator=g_) * 0.1 + 0.2          # mean
    st[1] = torch.rand(ci, generator=g_) + 0.5                 # invstd
    st[2] = st[1] * (torch.rand(ci, generator=g_) + 0.5)       # scale = gamma * invstd
    if neg_scales:  # (all-positive scales take the epilogue's monotone form of the window scan; a negative or zero gamma the scan)
        st[2][::5] *= -1.0
        st[2][3] = 0.0
    st[3] = torch.randn(ci, generator=g_) * 0.2 - st[0] * st[2]
    st = st.cuda()
    ws = torch.empty(n.call("spcl_bnrelu_bwd_workspace_bytes", N, H2, W2, ci) // 4, device="cuda")
    # reference path
    g_ref, _ = conv(n, dy_in, dtype, N, H, W, co, co, ci, wp_t, 0)
    dg0, db0 = torch.empty(ci, device="cuda"), torch.empty(ci, device="cuda")
    dy0 = torch.empty(N, H2, W2, ci, dtype=dtype, device="cuda")
    n.call("spcl_bnrelu_pool_backward", n.ptr(y2), None, n.ptr(g_ref), dtc, N, H2, W2, ci, ci, n.ptr(st[0]), n.ptr(st[1]),
           n.ptr(st[2]), n.ptr(st[3]), 1, n.ptr(ws), n.ptr(dg0), n.ptr(db0), n.ptr(dy0), n.stream())
    # fused path
    nt = n.call("spcl_conv_stat_rows", dtc, N, H, W, co, ci)
    g1 = torch.empty(N, H, W, ci, dtype=dtype, device="cuda")
    rows = torch.full((nt * 2 * ci,), float("nan"), device="cuda")
    n.call("spcl_conv3x3_dgrad_poolstats", n.ptr(dy_in), dtc, N, H, W, co, ci, n.ptr(wp_t), n.ptr(g1), n.ptr(y2), H2, W2,
           n.ptr(st[2]), n.ptr(st[3]), n.ptr(st[0]), n.ptr(rows), n.stream())
    assert torch.equal(g1, g_ref)
    assert not torch.isnan(rows).any()
    dg1, db1 = torch.empty(ci, device="cuda"), torch.empty(ci, device="cuda")
    dy1 = torch.empty(N, H2, W2, ci, dtype=dtype, device="cuda")
    n.call("spcl_bnrelu_pool_backward_rows", n.ptr(y2), n.ptr(g1), n.ptr(rows), nt, dtc, N, H2, W2, ci, ci, n.ptr(st[0]),
           n.ptr(st[1]), n.ptr(st[2]), n.ptr(st[3]), 1, n.ptr(ws), n.ptr(dg1), n.ptr(db1), n.ptr(dy1), n.stream())
    assert relerr(dg1, dg0) < 2e-5 and relerr(db1, db0) < 2e-5
    assert relerr(dy1.float(), dy0.float()) < 8e-3


def test_block_pack_writes_the_band_gemm_layout_only_where_that_kernel_runs():
    """spcl_conv_pack_weights_block_at: at an image size where the per-wave kernels run (28^2) the second half of a
    dual-layout buffer stays untouched and the first half equals the size-agnostic pack; at 32^2 (band GEMM kernel) and
    with H = W = 0 both halves are written, identical to spcl_conv_pack_weights_block."""
    n = _n()
    dtype, dtc = torch.bfloat16, n.dtype_code(torch.bfloat16)
    g = torch.Generator().manual_seed(3)
    wa, wb = torch.randn(128, 64, 3, 3, generator=g).cuda(), torch.randn(128, 128, 3, 3, generator=g).cuda()
    sizes = [n.call("spcl_conv_packed_elems", ci, co, kind, dtc) for (ci, co) in ((64, 128), (128, 128)) for kind in (0, 1)]

    def run(fn, *hw):
        bufs = [torch.full((s,), 7.0, dtype=dtype, device="cuda") for s in sizes]
        n.call(fn, n.ptr(wa), 64, 128, n.ptr(bufs[0]), n.ptr(bufs[1]), n.ptr(wb), 128, 128, n.ptr(bufs[2]), n.ptr(bufs[3]),
               dtc, *hw, n.stream())
        return bufs

    ref = run("spcl_conv_pack_weights_block")
    for b in ref:
        assert not torch.any(b == 7.0)  # every element written
    both = run("spcl_conv_pack_weights_block_at", 0, 0)
    big = run("spcl_conv_pack_weights_block_at", 32, 32)
    small = run("spcl_conv_pack_weights_block_at", 28, 28)
    for r, a, b, c in zip(ref, both, big, small):
        assert torch.equal(r, a) and torch.equal(r, b)
        half = r.numel() // 2
        assert torch.equal(c[:half], r[:half])
        assert torch.all(c[half:] == 7.0)


def test_copy_pair_is_two_copies():
    """spcl_copy_pair: both ranges copied by one launch, nothing beyond them touched"""
    n = _n()
    g = torch.Generator().manual_seed(1)
    a = torch.randn(3 * 224 * 224 + 4, generator=g).cuda()
    b = torch.randint(0, 4, (5 * 1000 + 2,), generator=g).cuda()  # int64
    da = torch.full((a.numel() + 8,), -7.0, device="cuda")
    db = torch.full((b.numel() + 4,), -7, dtype=torch.int64, device="cuda")
    n.call("spcl_copy_pair", da.data_ptr(), a.data_ptr(), a.numel() * 4, db.data_ptr(), b.data_ptr(), b.numel() * 8, n.stream())
    torch.cuda.synchronize()
    assert torch.equal(da[:a.numel()], a) and bool((da[a.numel():] == -7.0).all())
    assert torch.equal(db[:b.numel()], b) and bool((db[b.numel():] == -7).all())
    with pytest.raises(RuntimeError):
        n.call("spcl_copy_pair", da.data_ptr() + 4, a.data_ptr(), 16, db.data_ptr(), b.data_ptr(), 16, n.stream())


@pytest.mark.parametrize("dt", ["f32", "bf16"])
def test_multi_layer_pack_writes_what_the_single_layer_pack_writes(dt):
    """spcl_conv_pack_weights_multi (the step's one pack launch: 16 bytes per thread, one index decode per chunk) against
    spcl_conv_pack_weights (one element per thread) on the same weights: every byte of both layouts of every layer -- narrow
    layers, K padding (48 channels), the dual-layout bf16 layers (band-GEMM half included: H = W = 0), f32's exact + split
    halves -- and the zero-fill region the same launch serves."""
    n = _n()
    dtype = DT[dt]
    dtc = n.dtype_code(dtype)
    g = torch.Generator().manual_seed(7)
    layers = [(1, 16), (16, 16), (16, 32), (32, 48), (48, 64), (64, 64), (64, 128), (128, 256), (256, 128), (24, 40)]
    items, keep, outs = [], [], []
    for ci, co in layers:
        w = torch.randn(co, ci, 3, 3, generator=g).cuda()
        n0 = n.call("spcl_conv_packed_elems", ci, co, 0, dtc)
        n1 = n.call("spcl_conv_packed_elems", ci, co, 1, dtc)
        p0 = torch.full((n0,), 7.0, dtype=dtype, device="cuda")
        p1 = torch.full((n1,), 7.0, dtype=dtype, device="cuda")
        items.append(n.PackItem(w.data_ptr(), p0.data_ptr(), p1.data_ptr(), ci, co, 0, 0))
        keep.append(w)
        outs.append((w, p0, p1))
    zero = torch.full((1000,), 3.0, dtype=torch.float32, device="cuda")
    for i in range(0, len(items), n.PACK_MULTI_MAX):
        part = items[i:i + n.PACK_MULTI_MAX]
        arr = (n.PackItem * len(part))(*part)
        if i == 0:
            import ctypes
            n.call("spcl_conv_pack_weights_multi_zero", arr, len(part), dtc, None, 0, 0, 0, None, n.ptr(zero),
                   ctypes.c_size_t(zero.numel() * 4), n.stream())
        else:
            n.call("spcl_conv_pack_weights_multi", arr, len(part), dtc, n.stream())
    assert float(zero.abs().max()) == 0.0
    for w, p0, p1 in outs:
        r0, r1 = pack(n, w.cpu(), 0, dtype), pack(n, w.cpu(), 1, dtype)
        assert torch.equal(p0.view(torch.uint8), r0.view(torch.uint8)), tuple(w.shape)
        assert torch.equal(p1.view(torch.uint8), r1.view(torch.uint8)), tuple(w.shape)
