"""BatchNorm sums through fixed-point accumulator blocks (csrc/bn_acc.hpp; reference semi_seg/arch/unet.py:73,76:
nn.BatchNorm2d(momentum=0.1) in train mode and its autograd backward): the producing kernel's epilogue adds its tile's sums
with integer atomics, the next launch derives the coefficients in its prologue -- no finalize launch.  Through the C ABI, kernel
by kernel, against fp64 torch and against the rows + finalize path they replace; then the whole block against the oracle."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from tests.test_gpu_kernels import _n, conv, nhwc, pack, relerr, rnd, ru16  # noqa: E402

REPL, HI, LO = 8, 2.0 ** -10, 2.0 ** -60


def _block(n, cs):
    return torch.zeros(n.call("spcl_bn_acc_elems", cs), dtype=torch.int64, device="cuda")


def _totals(acc, cs):
    """decode a block on the host: [2][cs] float64 sums + the flag"""
    a = acc.cpu().numpy()
    w = a[:REPL * cs * 4].reshape(REPL, cs, 4).astype(object).sum(0)  # exact integer sums
    s1 = np.array([float(int(w[c, 0]) * HI + int(w[c, 1]) * LO) for c in range(cs)])
    s2 = np.array([float(int(w[c, 2]) * HI + int(w[c, 3]) * LO) for c in range(cs)])
    return s1, s2, int(a[REPL * cs * 4])


def _bn_desc(n, acc, gamma, beta, rm, rv, nbt, st, count, C, cs, momentum=0.1, eps=1e-5):
    return n.BnAcc(acc.data_ptr(), gamma.data_ptr(), beta.data_ptr(), rm.data_ptr() if rm is not None else None,
                   rv.data_ptr() if rv is not None else None, nbt.data_ptr() if nbt is not None else None, st.data_ptr(),
                   momentum, eps, float(count), C, cs)


ACC_SHAPES = [(2, 32, 64, 56, 56), (2, 64, 64, 56, 56), (3, 64, 128, 28, 28), (2, 128, 128, 28, 28), (4, 128, 256, 14, 14),
              (2, 256, 256, 14, 14), (1, 64, 64, 64, 64)]  # (the last: shifted last tiles)


@pytest.mark.parametrize("N,ci,co,H,W", ACC_SHAPES)
def test_forward_statistics_block_matches_fp64_and_the_rows(N, ci, co, H, W):
    """spcl_conv3x3_forward_acc(stats_acc): the block's totals are the batch sums of the convolution's output -- against fp64
    torch and against the per-tile Chan rows of spcl_conv3x3_forward -- the output itself bit-identical, two runs bit-identical"""
    n = _n()
    dtype, dtc = torch.bfloat16, 1
    assert n.call("spcl_conv_bn_acc_supported", dtc, N, H, W, ci, co, 0, 1)
    g = torch.Generator().manual_seed(ci + co + H)
    x = rnd(torch.randn(N, ci, H, W, generator=g) + 0.3, dtype)
    w = rnd(torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5), dtype)
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    xs, wp = nhwc(x, dtype), pack(n, w, 0, dtype)
    y0, st = conv(n, xs, dtype, N, H, W, ci, ci, co, wp, 0, stats=True)
    blocks = []
    for _ in range(2):
        acc = _block(n, co)
        y1 = torch.empty_like(y0)
        n.call("spcl_conv3x3_forward_acc", n.ptr(xs), dtc, N, H, W, ci, co, n.ptr(wp), None, None, None, n.ptr(y1), n.ptr(acc),
               None, n.stream())
        assert torch.equal(y1, y0)
        blocks.append(acc.clone())
    assert torch.equal(blocks[0], blocks[1])  # integer sums: the same bits whatever order the workgroups arrived in
    s1, s2, flag = _totals(blocks[0], co)
    assert flag == 0
    np.testing.assert_allclose(s1, ref.sum(dim=(0, 2, 3)).numpy(), rtol=2e-3, atol=2e-2)
    np.testing.assert_allclose(s2, (ref * ref).sum(dim=(0, 2, 3)).numpy(), rtol=2e-3)
    # the same f32 accumulators as the rows path: sums agree to f32 tile-sum rounding
    rows = st[:st.ntiles * 3 * co].view(st.ntiles, 3, co).double().cpu()
    r1 = (rows[:, 0] * rows[:, 1]).sum(0).numpy()
    r2 = (rows[:, 2] + rows[:, 0] * rows[:, 1] ** 2).sum(0).numpy()
    np.testing.assert_allclose(s1, r1, rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(s2, r2, rtol=1e-5)


@pytest.mark.parametrize("N,C,H,W,pool", [(2, 64, 56, 56, True), (3, 128, 28, 28, True), (2, 256, 14, 14, False),
                                          (2, 64, 28, 42, False)])
def test_block_consumers_equal_finalize_plus_apply(N, C, H, W, pool):
    """conv -> [block] -> conv with the coefficients derived in the prologue (MODE 5) -> [block] -> BN-apply + ReLU (+ pool)
    with the coefficients derived in ITS prologue, against the same three launches with spcl_bn_finalize in between: outputs
    within bf16 rounding of each other, mean / invstd / scale / shift and the running statistics to 1e-6."""
    n = _n()
    dtype, dtc, cs = torch.bfloat16, 1, C
    if not (n.call("spcl_conv_bn_acc_supported", dtc, N, H, W, C, C, 0, 1)
            and n.call("spcl_conv_bn_acc_supported", dtc, N, H, W, C, C, 1, 1)):
        pytest.skip("no accumulator form for this shape")
    g = torch.Generator().manual_seed(C + H)
    x = nhwc(rnd(torch.randn(N, C, H, W, generator=g), dtype), dtype)
    wa = pack(n, rnd(torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5), dtype), 0, dtype)
    wb = pack(n, rnd(torch.randn(C, C, 3, 3, generator=g) / (3 * C ** 0.5), dtype), 0, dtype)
    gam = [(torch.rand(C, generator=g) + 0.5).cuda() for _ in range(2)]
    bet = [(torch.randn(C, generator=g) * 0.2).cuda() for _ in range(2)]
    count = N * H * W

    def buffers():
        return ([torch.full((C,), 0.25, device="cuda") for _ in range(2)], [torch.full((C,), 0.75, device="cuda") for _ in range(2)],
                [torch.zeros((), dtype=torch.int64, device="cuda") for _ in range(2)])

    # ---- reference: rows + finalize
    rm0, rv0, nbt0 = buffers()
    ya0, sa = conv(n, x, dtype, N, H, W, cs, cs, cs, wa, 0, stats=True)
    st_a0 = torch.empty(4, cs, device="cuda")
    n.call("spcl_bn_finalize", n.ptr(sa), sa.ntiles, C, cs, n.ptr(gam[0]), n.ptr(bet[0]), ctypes.c_float(0.1),
           ctypes.c_float(1e-5), n.ptr(rm0[0]), n.ptr(rv0[0]), n.ptr(nbt0[0]), n.ptr(st_a0[0]), n.ptr(st_a0[1]), n.ptr(st_a0[2]),
           n.ptr(st_a0[3]), n.stream())
    yb0, sb = conv(n, ya0, dtype, N, H, W, cs, cs, cs, wb, 1, st_a0[2], st_a0[3], stats=True)
    st_b0 = torch.empty(4, cs, device="cuda")
    n.call("spcl_bn_finalize", n.ptr(sb), sb.ntiles, C, cs, n.ptr(gam[1]), n.ptr(bet[1]), ctypes.c_float(0.1),
           ctypes.c_float(1e-5), n.ptr(rm0[1]), n.ptr(rv0[1]), n.ptr(nbt0[1]), n.ptr(st_b0[0]), n.ptr(st_b0[1]), n.ptr(st_b0[2]),
           n.ptr(st_b0[3]), n.stream())
    act0 = torch.empty(N, H, W, cs, dtype=dtype, device="cuda")
    pool0 = torch.empty(N, H // 2, W // 2, cs, dtype=dtype, device="cuda") if pool else None
    n.call("spcl_bnrelu_pool_forward", n.ptr(yb0), dtc, N, H, W, cs, n.ptr(st_b0[2]), n.ptr(st_b0[3]), n.ptr(act0), n.ptr(pool0),
           n.stream())
    # ---- accumulator blocks
    rm1, rv1, nbt1 = buffers()
    acc_a, acc_b = _block(n, cs), _block(n, cs)
    st_a1, st_b1 = torch.full((4, cs), float("nan"), device="cuda"), torch.full((4, cs), float("nan"), device="cuda")
    ya1 = torch.empty_like(ya0)
    n.call("spcl_conv3x3_forward_acc", n.ptr(x), dtc, N, H, W, cs, cs, n.ptr(wa), None, None, None, n.ptr(ya1), n.ptr(acc_a), None,
           n.stream())
    assert torch.equal(ya1, ya0)
    bn_a = _bn_desc(n, acc_a, gam[0], bet[0], rm1[0], rv1[0], nbt1[0], st_a1, count, C, cs)
    yb1 = torch.empty_like(yb0)
    n.call("spcl_conv3x3_forward_acc", n.ptr(ya1), dtc, N, H, W, cs, cs, n.ptr(wb), ctypes.byref(bn_a), None, None, n.ptr(yb1),
           n.ptr(acc_b), None, n.stream())
    bn_b = _bn_desc(n, acc_b, gam[1], bet[1], rm1[1], rv1[1], nbt1[1], st_b1, count, C, cs)
    act1 = torch.empty_like(act0)
    pool1 = torch.empty_like(pool0) if pool else None
    n.call("spcl_bnrelu_pool_forward_acc", n.ptr(yb1), dtc, N, H, W, cs, ctypes.byref(bn_b), n.ptr(act1), n.ptr(pool1), n.stream())
    torch.cuda.synchronize()
    for a, b in ((st_a1, st_a0), (st_b1, st_b0)):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-5, atol=2e-6)
    for k in range(2):
        np.testing.assert_allclose(rm1[k].cpu().numpy(), rm0[k].cpu().numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(rv1[k].cpu().numpy(), rv0[k].cpu().numpy(), rtol=1e-5)
        assert int(nbt1[k]) == 1
    # coefficients that differ in their last bits move a few staged activations across a bf16 rounding boundary
    assert relerr(yb1.float(), yb0.float()) < 6e-3
    assert relerr(act1.float(), act0.float()) < 8e-3
    if pool:
        assert relerr(pool1.float(), pool0.float()) < 8e-3


@pytest.mark.parametrize("N,C,H,W", [(2, 64, 56, 56), (2, 128, 28, 28), (2, 256, 14, 14), (1, 64, 28, 42)])
def test_backward_block_from_the_dgrad_equals_rows_plus_finalize(N, C, H, W):
    """spcl_conv3x3_dgrad_bnstats_acc + spcl_bnrelu_backward_acc against spcl_conv3x3_dgrad_bnstats + spcl_bnrelu_backward_rows
    (their own finalize launch): g bit-identical, dgamma / dbeta to 2e-5, dy within bf16 rounding; deterministic"""
    n = _n()
    dtype, dtc, cs = torch.bfloat16, 1, C
    if not n.call("spcl_conv_dgrad_bnstats_acc_supported", dtc, N, H, W, cs, cs):
        pytest.skip("no accumulator form for this shape")
    g_ = torch.Generator().manual_seed(C + H)
    dy_in = nhwc(rnd(torch.randn(N, C, H, W, generator=g_) * 1e-3, dtype), dtype)  # (gradient-sized values)
    y2 = nhwc(rnd(torch.randn(N, C, H, W, generator=g_) * 1.3 + 0.2, dtype), dtype)
    wp_t = pack(n, torch.randn(C, C, 3, 3, generator=g_) / (3.0 * C ** 0.5), 1, dtype)
    st = torch.zeros(4, cs)
    st[0] = torch.randn(C, generator=g_) * 0.1 + 0.2
    st[1] = torch.rand(C, generator=g_) + 0.5
    st[2] = st[1] * (torch.rand(C, generator=g_) + 0.5)
    st[3] = torch.randn(C, generator=g_) * 0.2 - st[0] * st[2]
    st = st.cuda()
    ws = torch.empty(n.call("spcl_bnrelu_bwd_workspace_bytes", N, H, W, cs) // 4, device="cuda")
    nt = n.call("spcl_conv_stat_rows", dtc, N, H, W, cs, cs)
    g0 = torch.empty(N, H, W, cs, dtype=dtype, device="cuda")
    rows = torch.empty(nt * 2 * cs, device="cuda")
    n.call("spcl_conv3x3_dgrad_bnstats", n.ptr(dy_in), dtc, N, H, W, cs, cs, n.ptr(wp_t), n.ptr(g0), n.ptr(y2), n.ptr(st[2]),
           n.ptr(st[3]), n.ptr(st[0]), n.ptr(rows), n.stream())
    dg0, db0 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    dy0 = torch.empty(N, H, W, cs, dtype=dtype, device="cuda")
    n.call("spcl_bnrelu_backward_rows", n.ptr(y2), n.ptr(g0), None, n.ptr(rows), nt, dtc, N, H, W, C, cs, n.ptr(st[0]),
           n.ptr(st[1]), n.ptr(st[2]), n.ptr(st[3]), 1, n.ptr(ws), n.ptr(dg0), n.ptr(db0), n.ptr(dy0), None, n.stream())
    outs = []
    for _ in range(2):
        acc = _block(n, cs)
        g1 = torch.empty_like(g0)
        n.call("spcl_conv3x3_dgrad_bnstats_acc", n.ptr(dy_in), dtc, N, H, W, cs, cs, n.ptr(wp_t), n.ptr(g1), n.ptr(y2),
               n.ptr(st[2]), n.ptr(st[3]), n.ptr(st[0]), n.ptr(acc), n.stream())
        assert torch.equal(g1, g0)
        dg1, db1 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        dy1 = torch.empty_like(dy0)
        n.call("spcl_bnrelu_backward_acc", n.ptr(y2), n.ptr(g1), None, None, dtc, N, H, W, C, cs, n.ptr(st), 1, n.ptr(acc),
               n.ptr(dg1), n.ptr(db1), n.ptr(dy1), n.stream())
        outs.append((acc.clone(), dg1, db1, dy1))
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    _, dg1, db1, dy1 = outs[0]
    assert relerr(dg1, dg0) < 2e-5 and relerr(db1, db0) < 2e-5
    assert relerr(dy1.float(), dy0.float()) < 8e-3
    # the block holds the sums of the rows (both are sums of the same per-tile f32 values)
    s1, s2, flag = _totals(outs[0][0], cs)
    r = rows.view(nt, 2, cs).double().cpu()
    assert flag == 0
    np.testing.assert_allclose(s1, r[:, 0].sum(0).numpy(), rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(s2, r[:, 1].sum(0).numpy(), rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("N,ci,co,H2,W2", [(2, 32, 64, 112, 112), (2, 64, 128, 56, 56), (2, 128, 256, 28, 28)])
def test_pooled_backward_block_equals_rows_plus_finalize(N, ci, co, H2, W2):
    """spcl_conv3x3_dgrad_poolstats_acc + spcl_bnrelu_backward_acc(dpool) against the rows form"""
    n = _n()
    dtype, dtc = torch.bfloat16, 1
    H, W = H2 // 2, W2 // 2
    if not n.call("spcl_conv_dgrad_poolstats_acc_supported", dtc, N, H, W, co, ci, H2, W2):
        pytest.skip("no accumulator form for this shape")
    g_ = torch.Generator().manual_seed(ci + H2)
    dy_in = nhwc(rnd(torch.randn(N, co, H, W, generator=g_) * 1e-2, dtype), dtype)
    y2 = nhwc(rnd(torch.randn(N, ci, H2, W2, generator=g_) * 1.3 + 0.2, dtype), dtype)
    wp_t = pack(n, torch.randn(co, ci, 3, 3, generator=g_) / (3.0 * co ** 0.5), 1, dtype)
    st = torch.zeros(4, ci)
    st[0] = torch.randn(ci, generator=g_) * 0.1 + 0.2
    st[1] = torch.rand(ci, generator=g_) + 0.5
    st[2] = st[1] * (torch.rand(ci, generator=g_) + 0.5)
    st[2][::5] *= -1.0
    st[3] = torch.randn(ci, generator=g_) * 0.2 - st[0] * st[2]
    st = st.cuda()
    ws = torch.empty(n.call("spcl_bnrelu_bwd_workspace_bytes", N, H2, W2, ci) // 4, device="cuda")
    nt = n.call("spcl_conv_stat_rows", dtc, N, H, W, co, ci)
    g0 = torch.empty(N, H, W, ci, dtype=dtype, device="cuda")
    rows = torch.empty(nt * 2 * ci, device="cuda")
    n.call("spcl_conv3x3_dgrad_poolstats", n.ptr(dy_in), dtc, N, H, W, co, ci, n.ptr(wp_t), n.ptr(g0), n.ptr(y2), H2, W2,
           n.ptr(st[2]), n.ptr(st[3]), n.ptr(st[0]), n.ptr(rows), n.stream())
    dg0, db0 = torch.empty(ci, device="cuda"), torch.empty(ci, device="cuda")
    dy0 = torch.empty(N, H2, W2, ci, dtype=dtype, device="cuda")
    n.call("spcl_bnrelu_pool_backward_rows", n.ptr(y2), n.ptr(g0), n.ptr(rows), nt, dtc, N, H2, W2, ci, ci, n.ptr(st[0]),
           n.ptr(st[1]), n.ptr(st[2]), n.ptr(st[3]), 1, n.ptr(ws), n.ptr(dg0), n.ptr(db0), n.ptr(dy0), n.stream())
    acc = _block(n, ci)
    g1 = torch.empty_like(g0)
    n.call("spcl_conv3x3_dgrad_poolstats_acc", n.ptr(dy_in), dtc, N, H, W, co, ci, n.ptr(wp_t), n.ptr(g1), n.ptr(y2), H2, W2,
           n.ptr(st[2]), n.ptr(st[3]), n.ptr(st[0]), n.ptr(acc), n.stream())
    assert torch.equal(g1, g0)
    dg1, db1 = torch.empty(ci, device="cuda"), torch.empty(ci, device="cuda")
    dy1 = torch.empty_like(dy0)
    n.call("spcl_bnrelu_backward_acc", n.ptr(y2), None, n.ptr(g1), None, dtc, N, H2, W2, ci, ci, n.ptr(st), 1, n.ptr(acc),
           n.ptr(dg1), n.ptr(db1), n.ptr(dy1), n.stream())
    assert relerr(dg1, dg0) < 2e-5 and relerr(db1, db0) < 2e-5
    assert relerr(dy1.float(), dy0.float()) < 8e-3


@pytest.mark.parametrize("N,C,H,W", [(64, 256, 14, 14), (6, 128, 16, 16), (3, 64, 7, 9)])
def test_broadcast_gradient_backward_block_equals_the_three_launch_form(N, C, H, W):
    """spcl_bnrelu_backward_acc(dact_nc): the global-average-pool gradient (one value per image and channel) -- the reduction pass
    adds into the block, the apply pass derives its coefficients -- against spcl_bnrelu_backward_bcast (reduce, finalize, apply)"""
    n = _n()
    dtype, dtc, cs = torch.bfloat16, 1, C
    g_ = torch.Generator().manual_seed(C + N)
    y = nhwc(rnd(torch.randn(N, C, H, W, generator=g_) * 1.3 + 0.2, dtype), dtype)
    gnc = (torch.randn(N, C, generator=g_) * 1e-3).to(dtype).cuda()
    st = torch.zeros(4, cs)
    st[0] = torch.randn(C, generator=g_) * 0.1 + 0.2
    st[1] = torch.rand(C, generator=g_) + 0.5
    st[2] = st[1] * (torch.rand(C, generator=g_) + 0.5)
    st[3] = torch.randn(C, generator=g_) * 0.2 - st[0] * st[2]
    st = st.cuda()
    ws = torch.empty(n.call("spcl_bnrelu_bwd_workspace_bytes", N, H, W, cs) // 4, device="cuda")
    dg0, db0 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    dy0 = torch.empty(N, H, W, cs, dtype=dtype, device="cuda")
    n.call("spcl_bnrelu_backward_bcast", n.ptr(y), n.ptr(gnc), dtc, N, H, W, C, cs, n.ptr(st[0]), n.ptr(st[1]), n.ptr(st[2]),
           n.ptr(st[3]), 1, n.ptr(ws), n.ptr(dg0), n.ptr(db0), n.ptr(dy0), n.stream())
    outs = []
    for _ in range(2):
        acc = _block(n, cs)
        dg1, db1 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        dy1 = torch.empty_like(dy0)
        n.call("spcl_bnrelu_backward_acc", n.ptr(y), None, None, n.ptr(gnc), dtc, N, H, W, C, cs, n.ptr(st), 1, n.ptr(acc),
               n.ptr(dg1), n.ptr(db1), n.ptr(dy1), n.stream())
        outs.append((dg1, db1, dy1))
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    dg1, db1, dy1 = outs[0]
    assert relerr(dg1, dg0) < 2e-5 and relerr(db1, db0) < 2e-5
    assert relerr(dy1.float(), dy0.float()) < 8e-3


def test_out_of_range_sum_raises_the_flag_and_poisons_the_coefficients():
    """a tile sum outside the fixed-point range (or a NaN) must not pass silently: the block's flag goes up, the consumer's
    coefficients are NaN (the loss then turns NaN and the criterion raises, contrast_loss3.py:203-204)"""
    n = _n()
    dtype, dtc, N, C, H, W = torch.bfloat16, 1, 1, 64, 28, 28
    x = torch.zeros(N, C, H, W)
    x[0, 0, 5, 5] = float("nan")
    xs = nhwc(x, dtype)
    w = torch.zeros(C, C, 3, 3)
    w[:, :, 1, 1] = torch.eye(C)
    wp = pack(n, w, 0, dtype)
    acc = _block(n, C)
    y = torch.empty(N, H, W, C, dtype=dtype, device="cuda")
    n.call("spcl_conv3x3_forward_acc", n.ptr(xs), dtc, N, H, W, C, C, n.ptr(wp), None, None, None, n.ptr(y), n.ptr(acc), None,
           n.stream())
    assert _totals(acc, C)[2] > 0
    st = torch.zeros(4, C, device="cuda")
    gam, bet = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    bn = _bn_desc(n, acc, gam, bet, None, None, None, st, N * H * W, C, C)
    act = torch.empty_like(y)
    n.call("spcl_bnrelu_pool_forward_acc", n.ptr(y), dtc, N, H, W, C, ctypes.byref(bn), n.ptr(act), None, n.stream())
    assert torch.isnan(st).all()


def test_conv_block_with_and_without_blocks_and_bit_for_bit_determinism(monkeypatch):
    """functional.conv_block on the Conv3 .. Conv5 shapes with the accumulator blocks (default) and with the rows + finalize
    launches (SPCL_BN_ACC=0): activations, running statistics and every gradient agree to bf16 rounding; two runs with the
    blocks are bit-identical; and the block path really ran (no spcl_bn_finalize / spcl_bnrelu_backward_rows call)."""
    import spcl_amd  # noqa
    from spcl_amd import functional as Fh
    from spcl_amd.semi_seg.arch.unet import _ConvBlock
    for cin, cout, N, S in ((32, 64, 4, 56), (64, 128, 4, 28), (128, 256, 8, 14)):
        g = torch.Generator().manual_seed(cout)
        x0 = torch.randn(N, cin, S, S, generator=g)
        r = torch.randn(N, cout, S // 2, S // 2, generator=g) * 1e-2
        res = {}
        for tag, on in (("acc", True), ("acc2", True), ("rows", False)):
            monkeypatch.setattr(Fh, "_BN_ACC", on)
            torch.manual_seed(3)
            blk = _ConvBlock(cin, cout).cuda().train()
            blk._compute_dtype = torch.bfloat16
            calls = []
            real = Fh._n.call
            monkeypatch.setattr(Fh._n, "call", lambda name, *a: (calls.append(name), real(name, *a))[1])
            Fh.bn_acc_arena_begin(torch.device("cuda"))
            x = x0.cuda().to(torch.bfloat16).requires_grad_(True)
            _, pooled = Fh.conv_block(x, blk.conv[0].weight, blk.conv[1].weight, blk.conv[1].bias, blk.conv[3].weight,
                                      blk.conv[4].weight, blk.conv[4].bias, blk._cfg(False, True))
            (pooled.float() * r.cuda()).sum().backward()
            monkeypatch.setattr(Fh._n, "call", real)
            if on:
                assert "spcl_bn_finalize" not in calls and "spcl_conv3x3_forward_acc" in calls, calls
                assert "spcl_bnrelu_backward_acc" in calls and "spcl_bnrelu_backward_rows" not in calls, calls
            else:
                assert "spcl_bn_finalize" in calls and "spcl_conv3x3_forward_acc" not in calls
            res[tag] = dict(pooled=pooled.detach().float().clone(), dx=x.grad.float().clone(),
                            grads=[p.grad.clone() for p in blk.parameters()],
                            bufs=[b.clone() for b in blk.buffers()])
        a, a2, b = res["acc"], res["acc2"], res["rows"]
        assert torch.equal(a["pooled"], a2["pooled"]) and torch.equal(a["dx"], a2["dx"])
        assert all(torch.equal(p, q) for p, q in zip(a["grads"], a2["grads"]))
        assert all(torch.equal(p, q) for p, q in zip(a["bufs"], a2["bufs"]))
        assert relerr(a["pooled"], b["pooled"]) < 8e-3 and relerr(a["dx"], b["dx"]) < 2e-2
        for p, q in zip(a["grads"], b["grads"]):
            assert relerr(p, q) < 2e-2, (cin, cout)
        for p, q in zip(a["bufs"], b["bufs"]):
            np.testing.assert_allclose(p.float().cpu().numpy(), q.float().cpu().numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("N,C,cs,H,W,dtype", [(4, 256, 256, 14, 14, torch.bfloat16), (3, 128, 128, 16, 16, torch.bfloat16),
                                             (2, 40, 48, 9, 11, torch.bfloat16), (2, 64, 64, 14, 14, torch.float32),
                                             (1, 16, 16, 64, 64, torch.bfloat16)])
@pytest.mark.parametrize("from_block", [False, True])
def test_activation_writer_with_the_global_average_as_a_side_output(N, C, cs, H, W, dtype, from_block):
    """spcl_bnrelu_gap_forward: the activation is spcl_bnrelu_pool_forward's (resp. ..._acc's) bit for bit, gap[n][c] the mean
    over the image of the STORED activation (what the projector's AdaptiveAvgPool2d((1, 1)) of projectors/heads.py:78-92
    reads back), coefficients from arrays or derived from an accumulator block"""
    n = _n()
    dtc = 1 if dtype == torch.bfloat16 else 0
    assert n.call("spcl_bnrelu_gap_supported", dtc, H, W, C, cs)
    g = torch.Generator().manual_seed(C + H)
    y = torch.zeros(N, H, W, cs)
    y[..., :C] = torch.randn(N, H, W, C, generator=g) * 1.5 + 0.2
    y = y.to(dtype).cuda()
    gam, bet = torch.zeros(cs), torch.zeros(cs)
    gam[:C], bet[:C] = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    gam, bet = gam.cuda(), bet.cuda()
    count = N * H * W
    act0 = torch.empty_like(y)
    act1 = torch.empty_like(y)
    gap = torch.full((N, C), float("nan"), device="cuda")
    if from_block:
        if dtype != torch.bfloat16 or not n.call("spcl_conv_bn_acc_supported", dtc, N, H, W, cs, cs, 0, 1) or C != cs:
            pytest.skip("no accumulator producer for this shape")
        # a real producer: the statistics of a convolution's output, added to a block by its epilogue
        x = nhwc(rnd(torch.randn(N, cs, H, W, generator=g), dtype), dtype)
        wp = pack(n, rnd(torch.randn(cs, cs, 3, 3, generator=g) / (3 * cs ** 0.5), dtype), 0, dtype)
        accs = [_block(n, cs), _block(n, cs)]
        ys = [torch.empty(N, H, W, cs, dtype=dtype, device="cuda") for _ in range(2)]
        sts = [torch.empty(4, cs, device="cuda") for _ in range(2)]
        rms = [torch.zeros(cs, device="cuda") for _ in range(2)]
        rvs = [torch.ones(cs, device="cuda") for _ in range(2)]
        nbts = [torch.zeros((), dtype=torch.int64, device="cuda") for _ in range(2)]
        for k in range(2):
            n.call("spcl_conv3x3_forward_acc", n.ptr(x), dtc, N, H, W, cs, cs, n.ptr(wp), None, None, None, n.ptr(ys[k]),
                   n.ptr(accs[k]), None, n.stream())
        d0 = _bn_desc(n, accs[0], gam, bet, rms[0], rvs[0], nbts[0], sts[0], count, C, cs)
        d1 = _bn_desc(n, accs[1], gam, bet, rms[1], rvs[1], nbts[1], sts[1], count, C, cs)
        n.call("spcl_bnrelu_pool_forward_acc", n.ptr(ys[0]), dtc, N, H, W, cs, ctypes.byref(d0), n.ptr(act0), None, n.stream())
        n.call("spcl_bnrelu_gap_forward", n.ptr(ys[1]), dtc, N, H, W, C, cs, None, None, ctypes.byref(d1), n.ptr(act1), n.ptr(gap),
               n.stream())
        assert torch.equal(sts[0], sts[1]) and torch.equal(rms[0], rms[1]) and torch.equal(rvs[0], rvs[1]) and int(nbts[1]) == 1
    else:
        sc, sh = gam * 0.9, bet
        n.call("spcl_bnrelu_pool_forward", n.ptr(y), dtc, N, H, W, cs, n.ptr(sc), n.ptr(sh), n.ptr(act0), None, n.stream())
        n.call("spcl_bnrelu_gap_forward", n.ptr(y), dtc, N, H, W, C, cs, n.ptr(sc), n.ptr(sh), None, n.ptr(act1), n.ptr(gap),
               n.stream())
    torch.cuda.synchronize()
    assert torch.equal(act0, act1)
    ref = act1.double().mean(dim=(1, 2))[:, :C]
    np.testing.assert_allclose(gap.cpu().numpy(), ref.cpu().numpy(), rtol=2e-6, atol=1e-7)


def test_projector_takes_the_block_s_global_average_and_falls_back_when_it_is_stale(monkeypatch):
    """conv_block hangs the activation's global average on the tensor it returns; the projector then runs no pooling launch
    (spcl_proj_forward with feat == NULL) and gives the result of the ordinary path (another summation order: 1e-6); a tensor
    whose version counter moved since (an in-place write), or a slice of it, is pooled the ordinary way"""
    import spcl_amd.functional as F_hip
    from spcl_amd.semi_seg.arch.unet import _ConvBlock
    g = torch.Generator().manual_seed(3)
    blk = _ConvBlock(64, 128).cuda().train()
    blk._compute_dtype = torch.bfloat16
    blk.register_forward_hook(lambda m, i, o: None)  # (a feature tap, as semi_seg/arch/hook.py registers one: the block then leaves the average)
    x = torch.randn(4, 64, 14, 14, generator=g).cuda()
    w1, b1 = (torch.randn(32, 128, generator=g) * 0.1).cuda().requires_grad_(True), torch.zeros(32).cuda().requires_grad_(True)
    w2, b2 = (torch.randn(16, 32, generator=g) * 0.1).cuda().requires_grad_(True), torch.zeros(16).cuda().requires_grad_(True)
    rr = torch.randn(4, 16, generator=g).cuda()

    def run(gap_on, spoil=None):
        monkeypatch.setattr(F_hip, "_GAP", gap_on)
        blk.zero_grad(set_to_none=True)
        for p in (w1, b1, w2, b2):
            p.grad = None
        torch.manual_seed(0)
        act = blk(x)
        has = getattr(act, "_spcl_gap", None) is not None
        if spoil == "stale":  # (what an in-place write to the tensor does to its version counter)
            act._spcl_gap = (act._spcl_gap[0], act._version - 1)
        feat = act[:, :, :, :] if spoil == "slice" else act
        calls = []
        real = F_hip._n.call
        F_hip._n.call = lambda name, *a: (calls.append((name, a[0] if a else None)), real(name, *a))[1]
        try:
            z = F_hip.projector(feat, w1, b1, w2, b2, True)
        finally:
            F_hip._n.call = real
        (z * rr).sum().backward()  # (not z.square().sum(): that is N for normalised rows -- a zero gradient, pure rounding noise)
        pooled_given = [a0 for nm, a0 in calls if nm == "spcl_proj_forward"][0] is None
        return has, pooled_given, z.detach().clone(), w1.grad.clone(), blk.conv[0].weight.grad.clone()

    has1, given1, z1, g1, c1 = run(True)
    has0, given0, z0, g0, c0 = run(False)
    assert has1 and given1 and not has0 and not given0
    np.testing.assert_allclose(z1.cpu().numpy(), z0.cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(g1.cpu().numpy(), g0.cpu().numpy(), rtol=1e-4, atol=1e-6)
    assert relerr(c1.float(), c0.float()) < 2e-2  # (bf16 block: the pooled rows differ in their last bits, the gradients in bf16 roundings)
    for spoil in ("stale", "slice"):
        has, given, z, _, _ = run(True, spoil)
        assert has and not given, spoil
        assert torch.equal(z, z0), spoil


@pytest.mark.parametrize("form", ["lin", "strided", "pool", "pool+act", "up2"])
@pytest.mark.parametrize("N,C,H,W", [(2, 64, 56, 56), (3, 128, 28, 28), (2, 32, 112, 112), (1, 16, 30, 22)])
def test_backward_whose_reduction_pass_fills_the_block_equals_the_three_launch_form(form, N, C, H, W):
    """spcl_bnrelu_backward_fill_acc (reduction pass adds to the block, apply pass derives: no finalize launch) against
    spcl_bnrelu_pool_backward / _strided / spcl_bnrelu_backward_up2 (reduction pass, finalize, apply) for every form of
    incoming gradient the decoder produces (unet.py:85-97,193-230 backward): dy within bf16 rounding, dgamma / dbeta to 1e-5,
    two runs bit-identical"""
    n = _n()
    dtype, dtc, cs = torch.bfloat16, 1, C
    if form in ("pool", "pool+act") and (H % 2 or W % 2):
        pytest.skip("whole windows only in this test")
    g = torch.Generator().manual_seed(C + H + len(form))
    y = (torch.randn(N, H, W, cs, generator=g) * 1.2 + 0.1).to(dtype).cuda()
    gam, bet = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
    yf = y.double()
    mean, var = yf.mean(dim=(0, 1, 2)), yf.var(dim=(0, 1, 2), unbiased=False)
    st = torch.stack([mean, 1 / torch.sqrt(var + 1e-5), gam.double() / torch.sqrt(var + 1e-5),
                      bet.double() - mean * gam.double() / torch.sqrt(var + 1e-5)]).float().contiguous()
    dact = dpool = d_up = None
    stride = 0
    if form in ("lin", "pool+act"):
        dact = (torch.randn(N, H, W, cs, generator=g) * 1e-2).to(dtype).cuda()
    if form == "strided":
        wide = (torch.randn(N, H, W, 2 * cs, generator=g) * 1e-2).to(dtype).cuda()
        dact, stride = wide[..., cs:], 2 * cs  # the upper half of a concatenation's gradient, read in place
    if form in ("pool", "pool+act"):
        dpool = (torch.randn(N, H // 2, W // 2, cs, generator=g) * 1e-2).to(dtype).cuda()
    if form == "up2":
        d_up = (torch.randn(N, 2 * H, 2 * W, cs, generator=g) * 1e-2).to(dtype).cuda()
    ws = torch.empty(n.call("spcl_bnrelu_bwd_workspace_bytes", N, H, W, cs) // 4, device="cuda")
    dg0, db0, dy0 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda"), torch.empty_like(y)
    gs0 = torch.empty_like(y)
    if form == "up2":
        n.call("spcl_bnrelu_backward_up2", n.ptr(y), n.ptr(d_up), n.ptr(gs0), dtc, N, H, W, C, cs, n.ptr(st[0]), n.ptr(st[1]),
               n.ptr(st[2]), n.ptr(st[3]), 1, n.ptr(ws), n.ptr(dg0), n.ptr(db0), n.ptr(dy0), n.stream())
    elif form == "strided":
        n.call("spcl_bnrelu_pool_backward_strided", n.ptr(y), dact.data_ptr(), stride, None, dtc, N, H, W, C, cs, n.ptr(st[0]),
               n.ptr(st[1]), n.ptr(st[2]), n.ptr(st[3]), 1, n.ptr(ws), n.ptr(dg0), n.ptr(db0), n.ptr(dy0), n.stream())
    else:
        n.call("spcl_bnrelu_pool_backward", n.ptr(y), n.ptr(dact), n.ptr(dpool), dtc, N, H, W, C, cs, n.ptr(st[0]), n.ptr(st[1]),
               n.ptr(st[2]), n.ptr(st[3]), 1, n.ptr(ws), n.ptr(dg0), n.ptr(db0), n.ptr(dy0), n.stream())
    runs = []
    for _ in range(2):
        acc = _block(n, cs)
        dg1, db1, dy1 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda"), torch.empty_like(y)
        gs1 = torch.empty_like(y) if form == "up2" else None
        n.call("spcl_bnrelu_backward_fill_acc", n.ptr(y), gs1.data_ptr() if form == "up2" else (dact.data_ptr() if dact is not None else None),
               stride, n.ptr(dpool), n.ptr(d_up), dtc, N, H, W, C, cs, n.ptr(st), 1, n.ptr(acc), n.ptr(dg1), n.ptr(db1),
               n.ptr(dy1), n.stream())
        runs.append((dg1, db1, dy1, gs1))
    torch.cuda.synchronize()
    dg1, db1, dy1, gs1 = runs[0]
    assert all(torch.equal(a, b) for a, b in zip(runs[0][:3], runs[1][:3]))
    np.testing.assert_allclose(db1.cpu().numpy(), db0.cpu().numpy(), rtol=2e-5, atol=1e-6 * float(db0.abs().max()) + 1e-9)
    np.testing.assert_allclose(dg1.cpu().numpy(), dg0.cpu().numpy(), rtol=2e-5, atol=1e-6 * float(dg0.abs().max()) + 1e-9)
    assert relerr(dy1.float(), dy0.float()) < 4e-3
    if form == "up2":
        assert torch.equal(gs1, gs0)


@pytest.mark.parametrize("pool", [False, True])
@pytest.mark.parametrize("N,C,H,W,nrows", [(2, 32, 112, 112, 5000), (3, 16, 56, 56, 700), (1, 64, 28, 28, 9)])
def test_rows_into_a_block_equal_rows_plus_finalize(pool, N, C, H, W, nrows):
    """spcl_bnrelu_backward_rows_acc: per-tile rows (sum dz, sum dz (y - mean)) added to a block by one launch, the apply pass
    derives -- against spcl_bnrelu_backward_rows / spcl_bnrelu_pool_backward_rows (group + finalize + apply) on the same rows"""
    n = _n()
    dtype, dtc, cs = torch.bfloat16, 1, C
    g = torch.Generator().manual_seed(C + H + nrows)
    y = (torch.randn(N, H, W, cs, generator=g) * 1.2 + 0.1).to(dtype).cuda()
    gam, bet = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
    yf = y.double()
    mean, var = yf.mean(dim=(0, 1, 2)), yf.var(dim=(0, 1, 2), unbiased=False)
    st = torch.stack([mean, 1 / torch.sqrt(var + 1e-5), gam.double() / torch.sqrt(var + 1e-5),
                      bet.double() - mean * gam.double() / torch.sqrt(var + 1e-5)]).float().contiguous()
    rows = (torch.randn(nrows, 2, cs, generator=g) * 0.05).cuda()  # (any rows: the two forms must agree on them)
    dact = None if pool else (torch.randn(N, H, W, cs, generator=g) * 1e-2).to(dtype).cuda()
    dpool = (torch.randn(N, H // 2, W // 2, cs, generator=g) * 1e-2).to(dtype).cuda() if pool else None
    ws = torch.empty(n.call("spcl_bnrelu_bwd_workspace_bytes", N, H, W, cs) // 4, device="cuda")
    dg0, db0, dy0 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda"), torch.empty_like(y)
    if pool:
        n.call("spcl_bnrelu_pool_backward_rows", n.ptr(y), n.ptr(dpool), n.ptr(rows), nrows, dtc, N, H, W, C, cs, n.ptr(st[0]),
               n.ptr(st[1]), n.ptr(st[2]), n.ptr(st[3]), 1, n.ptr(ws), n.ptr(dg0), n.ptr(db0), n.ptr(dy0), n.stream())
    else:
        n.call("spcl_bnrelu_backward_rows", n.ptr(y), n.ptr(dact), None, n.ptr(rows), nrows, dtc, N, H, W, C, cs, n.ptr(st[0]),
               n.ptr(st[1]), n.ptr(st[2]), n.ptr(st[3]), 1, n.ptr(ws), n.ptr(dg0), n.ptr(db0), n.ptr(dy0), None, n.stream())
    acc = _block(n, cs)
    dg1, db1, dy1 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda"), torch.empty_like(y)
    n.call("spcl_bnrelu_backward_rows_acc", n.ptr(y), n.ptr(dact), n.ptr(dpool), n.ptr(rows), nrows, dtc, N, H, W, C, cs, n.ptr(st),
           1, n.ptr(acc), n.ptr(dg1), n.ptr(db1), n.ptr(dy1), n.stream())
    torch.cuda.synchronize()
    np.testing.assert_allclose(db1.cpu().numpy(), db0.cpu().numpy(), rtol=2e-5, atol=1e-6 * float(db0.abs().max()) + 1e-9)
    np.testing.assert_allclose(dg1.cpu().numpy(), dg0.cpu().numpy(), rtol=2e-5, atol=1e-6 * float(dg0.abs().max()) + 1e-9)
    assert relerr(dy1.float(), dy0.float()) < 4e-3
    s1, s2, flag = _totals(acc, cs)
    assert flag == 0
    # (a workgroup folds its rows in float before it adds: the block's totals are the rows' sums to float rounding)
    ref = rows[:, 0].double().sum(0).cpu().numpy()
    np.testing.assert_allclose(s1, ref, rtol=2e-5, atol=2e-6 * float(np.abs(ref).max()))


def test_second_backward_through_the_same_graph_falls_back_to_rows():
    """``loss.backward(retain_graph=True)`` twice: an accumulator block is zero only for the first backward after
    its forward -- both its producer (``_take_acc``) and the next block's input-gradient kernel (``PoolLink.acc``) take it once;
    the second backward takes the rows + finalize path and must give the same gradients (two correct reductions of the same
    sums: bf16 rounding of the intermediate gradients aside)."""
    import spcl_amd  # noqa
    from oracle import spcl_oracle as O
    from spcl_amd.semi_seg.arch import UNet
    net = UNet(input_dim=1, num_classes=4, max_channel=256)
    net.load_state_dict(O.init_unet_state(1, 4, 256, seed=3), strict=True)
    net.cuda().train()
    net.set_compute_dtype(torch.bfloat16)
    g = torch.Generator().manual_seed(8)
    x = torch.rand(4, 1, 224, 224, generator=g).cuda()
    r = torch.randn(4, 256, 14, 14, generator=g).cuda()
    loss = (net(x, until="Conv5").float() * r).sum()
    enc = [(k, p) for k, p in net.named_parameters() if k.startswith("_Conv")]
    loss.backward(retain_graph=True)
    first = {k: p.grad.detach().clone() for k, p in enc}
    for _, p in enc:
        p.grad = None
    loss.backward()  # (raised 'null pointer' from spcl_bnrelu_backward_acc before the fix)
    for k, p in enc:
        a, b = p.grad.double(), first[k].double()
        assert torch.isfinite(a).all() and float((a - b).norm() / b.norm().clamp_min(1e-30)) < 2e-2, k
