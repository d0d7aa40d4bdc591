"""torch.autograd.Function wrappers around the C ABI (native.py).  Device tensors in, device tensors out;
no host synchronisation anywhere in this file."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int

import torch

from . import native as _n

SP_NONE, SP_HARD, SP_SOFT = 0, 1, 2


# --------------------------------------------------------------------------------------------- gradient sinks
def take_grad_sink(param, needed: bool = True):
    """Where a parameter's gradient should be written: its ARMED sink (a contiguous fp32 view of a flat gradient bucket,
    installed by ddp.FlatParams.zero_grad) or None (allocate).  Claimed IN BACKWARD, at the point the kernel writes
    (a forward that never runs backward leaves the arming untouched).  A sink serves one backward per arming: the
    backward kernels write the gradient straight into the bucket and return a fresh view of it, which autograd adopts
    as ``param.grad`` without a copy; a second use of the same parameter in that step allocates normally and autograd
    adds it into the bucket in place.  ``ddp.GradBucket.gather`` disarms whatever was not claimed."""
    if not needed or param is None:
        return None
    sink = getattr(param, "_grad_sink", None)
    if sink is None or not getattr(param, "_grad_sink_armed", False):
        return None
    if sink.shape != param.shape or sink.dtype != torch.float32 or not sink.is_contiguous() or sink.device != param.device:
        return None
    param._grad_sink_armed = False
    return sink


def _grad_buffer(sink, shape, dev):
    return sink.view(shape) if sink is not None else torch.empty(shape, dtype=torch.float32, device=dev)


# --------------------------------------------------------------------------------------------- contrastive loss
class SupConState:
    """Device-side results of one loss evaluation (kept for backward and for the lazily materialised taps)."""
    __slots__ = ("n", "d", "t", "sp_mode", "gamma", "labels", "mask", "ws", "out")


class _SupConFn(torch.autograd.Function):
    """``stacked``: z1 is the whole [2n, d] projection (view 1 rows, then view 2 rows) and z2 is None; the gradient
    comes back as one [2n, d] tensor (no chunk / cat copies around the loss)."""

    @staticmethod
    def forward(ctx, z1, z2, labels, mask, t, sp_mode, gamma, correct_grad, state: SupConState, raw=False):
        stacked = z2 is None
        _n.require_gpu(z1, labels, mask) if stacked else _n.require_gpu(z1, z2, labels, mask)
        z1c = z1.detach().contiguous().float()
        if stacked:
            assert z1c.dim() == 2 and z1c.shape[0] % 2 == 0, z1c.shape
            n, d = z1c.shape[0] // 2, z1c.shape[1]
            z2c = z1c[n:]
        else:
            z2c = z2.detach().contiguous().float()
            n, d = z1c.shape
        nbytes = _n.call("spcl_supcon_workspace_bytes", n, d)
        if nbytes == 0:
            raise RuntimeError(f"supcon: unsupported shape n={n} d={d} (d must be <= 4096)")
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=z1.device)
        out = torch.empty(8, dtype=torch.float32, device=z1.device)  # the kernel writes loss, rho, kappa, norm defect
        if raw:
            # the rows BEFORE F.normalize: the launch normalises them itself and its unit-gradient block is d loss / d raw
            # rows (spcl_supcon_forward_rows; supcon_loss has checked that the shape is supported and there is no mask)
            gam = (c_float * 1)(float(gamma))
            _n.call("spcl_supcon_forward_rows", 1, _n.ptr(z1c), _n.ptr(z2c), 0, _n.ptr(labels), n, d, c_float(t), sp_mode,
                    gam, int(bool(correct_grad)), _n.ptr(ws), 0, _n.ptr(out), _n.stream())
        else:
            _n.call("spcl_supcon_forward", _n.ptr(z1c), _n.ptr(z2c), _n.ptr(labels), _n.ptr(mask), n, d, c_float(t),
                    sp_mode, c_float(gamma), int(bool(correct_grad)), _n.ptr(ws), _n.ptr(out), _n.stream())
        state.n, state.d, state.t, state.sp_mode, state.gamma = n, d, t, sp_mode, gamma
        state.labels, state.mask, state.ws, state.out = labels, mask, ws, out
        ctx.state = state
        ctx.stacked = stacked
        ctx.in_dtypes = (z1.dtype, z1.dtype if stacked else z2.dtype)
        # a LEAF input keeps the gradient it is handed as its .grad (AccumulateGrad steals the buffer): it must then own it
        ctx.leaf_input = bool(z1.is_leaf)
        return out[0]

    @staticmethod
    def backward(ctx, grad_out):
        s = ctx.state
        dev = s.ws.device
        if ctx.stacked and ctx.in_dtypes[0] == torch.float32 and is_unit_gradient(grad_out):
            # the epocher's own ``backward(gradient=ones)``: the forward already left dLoss/dP for a unit gradient in its
            # workspace (training sizes) -- that block IS the gradient, no scaling launch
            off, pitch = ctypes.c_size_t(0), ctypes.c_int(0)
            if (_n.call("spcl_supcon_unit_gradient_block", s.n, s.d, ctypes.byref(off), ctypes.byref(pitch))
                    and pitch.value == s.d):
                dz = s.ws[off.value:off.value + 2 * s.n * s.d].view(2 * s.n, s.d)
                if ctx.leaf_input:  # (in-place operations on that .grad would write into the workspace the taps read)
                    dz = dz.clone()
                return dz, None, None, None, None, None, None, None, None, None
        dz = torch.empty(2 * s.n, s.d, dtype=torch.float32, device=dev)
        dz1, dz2 = dz[:s.n], dz[s.n:]
        wsb = torch.empty(_n.call("spcl_supcon_bwd_workspace_bytes", s.n, s.d) // 4, dtype=torch.float32, device=dev)
        go = grad_out.detach().reshape(1).float().contiguous()
        _n.call("spcl_supcon_backward", _n.ptr(s.labels), _n.ptr(s.mask), s.n, s.d, c_float(s.t), s.sp_mode,
                c_float(s.gamma), _n.ptr(s.ws), _n.ptr(wsb), _n.ptr(s.out), _n.ptr(go), _n.ptr(dz1), _n.ptr(dz2),
                _n.stream())
        if ctx.stacked:
            return dz.to(ctx.in_dtypes[0]), None, None, None, None, None, None, None, None, None
        return dz1.to(ctx.in_dtypes[0]), dz2.to(ctx.in_dtypes[1]), None, None, None, None, None, None, None, None


def supcon_rows_supported(n, d, mask=None):
    """can the loss launch take the rows BEFORE F.normalize (``normalize_inputs``)?  training sizes, no explicit mask"""
    return mask is None and bool(_n.call("spcl_supcon_rows_supported", int(n), int(d)))


def supcon_loss(z1, z2, labels=None, mask=None, *, t=0.07, sp_mode=SP_NONE, gamma=1e6, correct_grad=False,
                state: SupConState = None, normalize_inputs=False):
    """loss (0-dim tensor with grad_fn).  ``state`` receives the device-side statistics (rho = state.out[1]).
    ``normalize_inputs``: z1 / z2 are rows BEFORE ``F.normalize(dim=1)``; the loss is that of the normalised rows and
    the gradient the one w.r.t. the given rows -- inside the loss launch where its schedule allows
    (``supcon_rows_supported``), by the row-normalisation kernels in front of it otherwise."""
    if state is None:
        state = SupConState()
    raw = False
    if normalize_inputs:
        n = z1.shape[0] // 2 if z2 is None else z1.shape[0]
        if z1.dtype == torch.float32 and supcon_rows_supported(n, z1.shape[1], mask):
            raw = True
        else:
            z1 = l2norm_rows(z1)
            z2 = l2norm_rows(z2) if z2 is not None else None
    return _SupConFn.apply(z1, z2, labels, mask, float(t), int(sp_mode), float(gamma), bool(correct_grad), state, raw)


class _SupConHeadsFn(torch.autograd.Function):
    """K (2..4) losses of one shape in the launches of one (spcl_supcon_forward_heads / _backward_heads): the K meta-label
    hooks on one feature, each with its own labels and age parameter.  Inputs after the fixed ones: the K stacked
    [2n, d] projections; output: a [K] tensor of losses.  Per head the arithmetic (and the workspace / result block the
    ``states`` receive) is exactly that of ``_SupConFn``."""

    @staticmethod
    def forward(ctx, labels, t, sp_mode, gammas, correct_grad, states, raw, *zs):
        K = len(zs)
        _n.require_gpu(*zs)
        n2, d = zs[0].shape
        assert n2 % 2 == 0 and all(z.shape == zs[0].shape for z in zs), [tuple(z.shape) for z in zs]
        n = n2 // 2
        dev = zs[0].device
        step = n2 * d * 4
        if all(z.dtype == torch.float32 and z.is_contiguous() and z.data_ptr() == zs[0].data_ptr() + k * step
               for k, z in enumerate(zs)):
            zall = zs[0].detach()  # the batched projection wrote its heads back to back: no copy
        else:
            zall = torch.stack([z.detach().float() for z in zs])
        nbytes = _n.call("spcl_supcon_workspace_bytes", n, d)
        if nbytes == 0:
            raise RuntimeError(f"supcon: unsupported shape n={n} d={d} (d must be <= 4096)")
        ws_stride = (nbytes // 4 + 63) // 64 * 64
        ws = torch.empty(K * ws_stride, dtype=torch.float32, device=dev)
        out = torch.empty(K, 8, dtype=torch.float32, device=dev)
        gam = (c_float * K)(*[float(g) for g in gammas])
        base = zall.data_ptr()
        _n.call("spcl_supcon_forward_rows" if raw else "spcl_supcon_forward_heads", K, base, base + n * d * 4, n2 * d,
                _n.ptr(labels), n, d, c_float(t), sp_mode, gam, int(bool(correct_grad)), _n.ptr(ws), ws_stride, _n.ptr(out),
                _n.stream())
        for k, st in enumerate(states):
            st.n, st.d, st.t, st.sp_mode, st.gamma = n, d, t, sp_mode, float(gammas[k])
            st.labels, st.mask = (labels[k] if labels is not None else None), None
            st.ws, st.out = ws[k * ws_stride:(k + 1) * ws_stride], out[k]
        ctx.keep = (labels, ws, out, zall)
        ctx.meta = (K, n, d, t, sp_mode, tuple(float(g) for g in gammas), ws_stride, [z.dtype for z in zs])
        # K separate 0-dim losses (views of the result block: no copy; a [K] tensor that the hooks then unbind costs a
        # stacking launch in its backward)
        return tuple(out[k, 0] for k in range(K))

    @staticmethod
    def backward(ctx, *grads_out):
        K, n, d, t, sp_mode, gammas, ws_stride, dtypes = ctx.meta
        labels, ws, out, _ = ctx.keep
        dev = ws.device
        dz = torch.empty(K, 2 * n, d, dtype=torch.float32, device=dev)
        wsb_stride = (_n.call("spcl_supcon_bwd_workspace_bytes", n, d) // 4 + 63) // 64 * 64
        wsb = torch.empty(K * wsb_stride, dtype=torch.float32, device=dev)
        if all(g is not None and is_unit_gradient(g) for g in grads_out):
            go = _ones_k(K, dev)  # every head's upstream gradient is the epocher's registered 1: a cached vector of ones
        else:
            go = torch.stack([torch.zeros((), dtype=torch.float32, device=dev) if g is None else g.detach().float().reshape(())
                              for g in grads_out])
        gam = (c_float * K)(*gammas)
        base = dz.data_ptr()
        _n.call("spcl_supcon_backward_heads", K, _n.ptr(labels), n, d, c_float(t), sp_mode, gam, _n.ptr(ws), ws_stride,
                _n.ptr(wsb), wsb_stride, _n.ptr(out), _n.ptr(go), base, base + n * d * 4, 2 * n * d, _n.stream())
        return (None, None, None, None, None, None, None) + tuple(dz[k].to(dtypes[k]) for k in range(K))


_ONES_K = {}


def _ones_k(K, dev):
    key = (K, str(dev))
    t = _ONES_K.get(key)
    if t is None:
        t = _ONES_K[key] = torch.ones(K, dtype=torch.float32, device=dev)
    return t


def supcon_loss_heads(zs, labels=None, *, t=0.07, sp_mode=SP_NONE, gammas=None, correct_grad=False, states=None,
                      normalize_inputs=False):
    """K 0-dim losses (a tuple) of the K stacked [2n, d] projections ``zs`` (2 <= K <= 4); ``labels``: [K, n] float tensor or
    None; ``gammas``: K floats; ``states``: K SupConState objects that receive each head's device-side statistics.
    ``normalize_inputs``: as in ``supcon_loss``."""
    K = len(zs)
    if states is None:
        states = [SupConState() for _ in range(K)]
    if gammas is None:
        gammas = [1e6] * K
    raw = False
    if normalize_inputs:
        if all(z.dtype == torch.float32 for z in zs) and supcon_rows_supported(zs[0].shape[0] // 2, zs[0].shape[1]):
            raw = True
        else:
            zs = [l2norm_rows(z) for z in zs]
    return _SupConHeadsFn.apply(labels, float(t), int(sp_mode), tuple(gammas), bool(correct_grad), states, raw, *zs)


class _SupConXposFn(torch.autograd.Function):
    """SupConLoss1(exclude_other_pos=True), contrast_loss3.py:97-100 (csrc/supcon_xpos.hip)"""

    @staticmethod
    def forward(ctx, z1, z2, labels, mask, t, out):
        _n.require_gpu(z1, z2, labels, mask)
        z1c, z2c = z1.detach().contiguous().float(), z2.detach().contiguous().float()
        n, d = z1c.shape
        nbytes = _n.call("spcl_supcon_xpos_workspace_bytes", n, d)
        if nbytes == 0:
            raise RuntimeError(f"supcon (exclude_other_pos): unsupported shape n={n} d={d} (n <= 4096)")
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=z1.device)
        _n.call("spcl_supcon_xpos_forward", _n.ptr(z1c), _n.ptr(z2c), _n.ptr(labels), _n.ptr(mask), n, d, c_float(t),
                _n.ptr(ws), _n.ptr(out), _n.stream())
        ctx.save_for_backward(z1c, z2c, ws)
        ctx.meta = (n, d, t, z1.dtype, z2.dtype)
        return out[0]

    @staticmethod
    def backward(ctx, grad_out):
        z1c, z2c, ws = ctx.saved_tensors
        n, d, t, dt1, dt2 = ctx.meta
        dz1, dz2 = torch.empty_like(z1c), torch.empty_like(z2c)
        go = grad_out.detach().reshape(1).float().contiguous()
        _n.call("spcl_supcon_xpos_backward", _n.ptr(z1c), _n.ptr(z2c), n, d, c_float(t), _n.ptr(ws), _n.ptr(go),
                _n.ptr(dz1), _n.ptr(dz2), _n.stream())
        return dz1.to(dt1), dz2.to(dt2), None, None, None, None


def supcon_loss_exclude_other_pos(z1, z2, labels=None, mask=None, *, t=0.07, out=None):
    """-> loss (0-dim tensor with grad_fn); ``out`` (8 floats on the device) receives loss / norm defect"""
    if out is None:
        out = torch.empty(8, dtype=torch.float32, device=z1.device)
    return _SupConXposFn.apply(z1, z2, labels, mask, float(t), out)


def stacked_halves(a: torch.Tensor, b: torch.Tensor):
    """the tensor whose first / second half of rows a and b are (``torch.chunk(z, 2)`` outputs), else None"""
    base = a._base
    if (base is None or base is not b._base or a.dim() != 2 or a.shape != b.shape or base.dim() != 2
            or not base.is_contiguous() or base.shape[0] != 2 * a.shape[0] or base.shape[1] != a.shape[1]
            or a.stride() != base.stride() or b.stride() != base.stride()
            or a.storage_offset() != base.storage_offset() or b.storage_offset() != a.storage_offset() + a.numel()):
        return None
    return base


def supcon_materialize(state: SupConState, want=("sim_logits", "sim_exp", "pos_mask", "neg_mask", "sp_mask")):
    n2 = 2 * state.n
    dev = state.ws.device
    bufs = {k: (torch.empty(n2, n2, dtype=torch.float32, device=dev) if k in want else None)
            for k in ("sim_logits", "sim_exp", "pos_mask", "neg_mask", "sp_mask")}
    _n.call("spcl_supcon_materialize", _n.ptr(state.labels), _n.ptr(state.mask), state.n, state.d, c_float(state.t),
            state.sp_mode, c_float(state.gamma), _n.ptr(state.ws), _n.ptr(bufs["sim_logits"]),
            _n.ptr(bufs["sim_exp"]), _n.ptr(bufs["pos_mask"]), _n.ptr(bufs["neg_mask"]), _n.ptr(bufs["sp_mask"]),
            _n.stream())
    return bufs


# --------------------------------------------------------------------------------------------- NHWC plumbing
def as_nhwc(t: torch.Tensor):
    """Return (storage, Cs) where storage is a contiguous [N,H,W,Cs] tensor aliasing (or copying) the logical
    NCHW tensor ``t``.  Zero-copy when ``t`` already is channels-last (optionally channel-padded)."""
    assert t.dim() == 4
    N, C, H, W = t.shape
    sN, sC, sH, sW = t.stride()
    if sC == 1 and sW >= C and sH == W * sW and sN == H * W * sW and t.dtype in (torch.float32, torch.bfloat16):
        cs = sW
        if cs == C:
            return t.permute(0, 2, 3, 1), cs
        base = torch.as_strided(t, (N, H, W, cs), (sN, sH, sW, 1))
        return base, cs
    if t.dtype not in (torch.float32, torch.bfloat16):
        t = t.float()
    t = t.contiguous(memory_format=torch.channels_last)
    return t.permute(0, 2, 3, 1), C


def nhwc_to_logical(storage: torch.Tensor, C: int):
    """[N,H,W,Cs] storage -> logical NCHW view with C channels (channels-last strides, no copy)."""
    v = storage.permute(0, 3, 1, 2)
    return v if storage.shape[3] == C else v[:, :C]


# --------------------------------------------------------------------------------------------- projector
class _ProjectorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, w1, b1, w2, b2, normalize):
        _n.require_gpu(feat, w1, b1, w2, b2)
        x, cs = as_nhwc(feat.detach())
        N, H, W, _ = x.shape
        C = feat.shape[1]
        dev = feat.device
        mlp = w2 is not None
        hid = w1.shape[0] if mlp else 0
        out_dim = w2.shape[0] if mlp else w1.shape[0]
        w1c, b1c = w1.detach().contiguous().float(), b1.detach().contiguous().float()
        w2c = w2.detach().contiguous().float() if mlp else None
        b2c = b2.detach().contiguous().float() if mlp else None
        gap = _gap_of(feat, N, C) if H * W > 1 else None  # (the rows the feature map's producer left: no pooling launch)
        pooled = gap if gap is not None else torch.empty(N, C, dtype=torch.float32, device=dev)
        pre = torch.empty(N, hid, dtype=torch.float32, device=dev) if mlp else None
        o = torch.empty(N, out_dim, dtype=torch.float32, device=dev)
        z = torch.empty(N, out_dim, dtype=torch.float32, device=dev) if normalize else o  # (no copy without normalisation)
        _n.call("spcl_proj_forward", None if gap is not None else _n.ptr(x), _n.dtype_code(x.dtype), N, H * W, C, cs, _n.ptr(w1c), _n.ptr(b1c),
                _n.ptr(w2c), _n.ptr(b2c), hid, out_dim, int(bool(normalize)), _n.ptr(pooled), _n.ptr(pre), _n.ptr(o),
                _n.ptr(z), _n.stream())
        # (without normalisation the output IS o: the backward does not read it then, and an output must not be saved raw)
        ctx.save_for_backward(w1c, w2c, pooled, pre, o if normalize else None)
        ctx.meta = (N, H, W, C, cs, hid, out_dim, bool(normalize), x.dtype, feat.dtype)
        ctx.params = (w1, b1, w2, b2)  # their gradient sinks are claimed in backward, where the kernels write
        return z

    @staticmethod
    def backward(ctx, dz):
        w1c, w2c, pooled, pre, o = ctx.saved_tensors
        N, H, W, C, cs, hid, out_dim, normalize, xdt, fdt = ctx.meta
        dev = dz.device
        mlp = hid > 0
        dzc = dz.detach().contiguous().float()
        ng = ctx.needs_input_grad
        sk = tuple(take_grad_sink(p, ng[i + 1]) for i, p in enumerate(ctx.params))
        dw1 = _grad_buffer(sk[0], w1c.shape, dev)
        db1 = _grad_buffer(sk[1], (w1c.shape[0],), dev)
        dw2 = _grad_buffer(sk[2], w2c.shape, dev) if mlp else None
        db2 = _grad_buffer(sk[3], (out_dim,), dev) if mlp else None
        scratch = torch.empty(N * (out_dim + hid + C), dtype=torch.float32, device=dev)
        need_dfeat = ctx.needs_input_grad[0]
        if need_dfeat and _BCAST and cs == C and H * W > 1 and xdt == fdt:
            # the gradient of the global average pool as what it is -- one value per (image, channel) -- behind a stride-0
            # view: the block it came from reads the [N, C] form (broadcast_rows), anything else sees an ordinary tensor
            dnc = torch.empty(N, C, dtype=xdt, device=dev)
            one = lambda t: _n.ptr_array([t])  # noqa: E731
            _n.call("spcl_proj_heads_backward_pooled", 1, one(dzc), _n.dtype_code(xdt), N, H * W, C, cs, one(w1c), one(w2c),
                    hid, out_dim, int(normalize), _n.ptr(pooled), one(pre), one(o), one(dw1), one(db1), one(dw2), one(db2),
                    _n.ptr(scratch), _n.ptr(dnc), _n.stream())
            return dnc.view(N, C, 1, 1).expand(N, C, H, W), dw1, db1, dw2, db2, None
        dfeat = torch.empty(N, H, W, cs, dtype=xdt, device=dev) if need_dfeat else None
        _n.call("spcl_proj_backward", _n.ptr(dzc), _n.dtype_code(xdt), N, H * W, C, cs, _n.ptr(w1c), _n.ptr(w2c), hid,
                out_dim, int(normalize), _n.ptr(pooled), _n.ptr(pre), _n.ptr(o), _n.ptr(dw1), _n.ptr(db1),
                _n.ptr(dw2), _n.ptr(db2), _n.ptr(scratch), _n.ptr(dfeat), _n.stream())
        gfeat = None
        if need_dfeat:
            gfeat = nhwc_to_logical(dfeat, C)
            if gfeat.dtype != fdt:
                gfeat = gfeat.to(fdt)
        return gfeat, dw1, db1, dw2, db2, None


def projector(feat, w1, b1, w2=None, b2=None, normalize=True):
    return _ProjectorFn.apply(feat, w1, b1, w2, b2, normalize)


class _ProjectorHeadsFn(torch.autograd.Function):
    """K <= 4 MLP heads of identical shape on the SAME feature (several meta-label hooks on one encoder tap): the feature is
    pooled once and every layer of all heads is one launch (spcl_proj_heads_forward / _backward).  Inputs: feat, normalize,
    then (w1, b1, w2, b2) per head; outputs: one z [N, out] per head."""

    @staticmethod
    def forward(ctx, feat, normalize, *params):
        K = len(params) // 4
        heads = [params[4 * k:4 * k + 4] for k in range(K)]
        _n.require_gpu(feat, *params)
        x, cs = as_nhwc(feat.detach())
        N, H, W, _ = x.shape
        C = feat.shape[1]
        dev = feat.device
        hid, out_dim = heads[0][0].shape[0], heads[0][2].shape[0]
        cont = [[t.detach().contiguous().float() for t in h] for h in heads]
        gap = _gap_of(feat, N, C) if H * W > 1 else None
        pooled = gap if gap is not None else torch.empty(N, C, dtype=torch.float32, device=dev)
        pre = torch.empty(K, N, hid, dtype=torch.float32, device=dev)
        o = torch.empty(K, N, out_dim, dtype=torch.float32, device=dev)
        # back to back: the batched loss reads them in place (without normalisation the heads' rows are the output: no copy)
        z = list((torch.empty(K, N, out_dim, dtype=torch.float32, device=dev) if normalize else o).unbind(0))
        col = lambda i: _n.ptr_array([c[i] for c in cont])  # noqa: E731
        _n.call("spcl_proj_heads_forward", K, None if gap is not None else _n.ptr(x), _n.dtype_code(x.dtype), N, H * W, C, cs, col(0), col(1), col(2),
                col(3), hid, out_dim, int(bool(normalize)), _n.ptr(pooled), _n.ptr_array(list(pre)), _n.ptr_array(list(o)),
                _n.ptr_array(z), _n.stream())
        ctx.save_for_backward(pooled, pre, o if normalize else None, *[c[0] for c in cont], *[c[2] for c in cont])
        ctx.meta = (K, N, H, W, C, cs, hid, out_dim, bool(normalize), x.dtype, feat.dtype)
        ctx.params = params
        return tuple(z)

    @staticmethod
    def backward(ctx, *dzs):
        K, N, H, W, C, cs, hid, out_dim, normalize, xdt, fdt = ctx.meta
        saved = ctx.saved_tensors
        pooled, pre, o = saved[:3]
        o = [None] * K if o is None else list(o)  # (without normalisation the backward does not read the rows)
        w1s, w2s = saved[3:3 + K], saved[3 + K:3 + 2 * K]
        dev = pooled.device
        dzc = [torch.zeros(N, out_dim, dtype=torch.float32, device=dev) if g is None else g.detach().contiguous().float()
               for g in dzs]
        ng = ctx.needs_input_grad
        sk = [take_grad_sink(p, ng[i + 2]) for i, p in enumerate(ctx.params)]
        shapes = ((hid, C), (hid,), (out_dim, hid), (out_dim,))
        grads = [_grad_buffer(sk[i], shapes[i % 4], dev) for i in range(4 * K)]
        scratch = torch.empty(K * N * (out_dim + hid) + N * C, dtype=torch.float32, device=dev)
        need_dfeat = ng[0]
        col = lambda i: _n.ptr_array([grads[4 * k + i] for k in range(K)])  # noqa: E731
        if need_dfeat and _BCAST and cs == C and H * W > 1 and xdt == fdt:  # (see _ProjectorFn.backward)
            dnc = torch.empty(N, C, dtype=xdt, device=dev)
            _n.call("spcl_proj_heads_backward_pooled", K, _n.ptr_array(dzc), _n.dtype_code(xdt), N, H * W, C, cs,
                    _n.ptr_array(list(w1s)), _n.ptr_array(list(w2s)), hid, out_dim, int(normalize), _n.ptr(pooled),
                    _n.ptr_array(list(pre)), _n.ptr_array(o), col(0), col(1), col(2), col(3), _n.ptr(scratch),
                    _n.ptr(dnc), _n.stream())
            return (dnc.view(N, C, 1, 1).expand(N, C, H, W), None) + tuple(grads)
        dfeat = torch.empty(N, H, W, cs, dtype=xdt, device=dev) if need_dfeat else None
        _n.call("spcl_proj_heads_backward", K, _n.ptr_array(dzc), _n.dtype_code(xdt), N, H * W, C, cs, _n.ptr_array(list(w1s)),
                _n.ptr_array(list(w2s)), hid, out_dim, int(normalize), _n.ptr(pooled), _n.ptr_array(list(pre)),
                _n.ptr_array(o), col(0), col(1), col(2), col(3), _n.ptr(scratch), _n.ptr(dfeat), _n.stream())
        gfeat = None
        if need_dfeat:
            gfeat = nhwc_to_logical(dfeat, C)
            if gfeat.dtype != fdt:
                gfeat = gfeat.to(fdt)
        return (gfeat, None) + tuple(grads)


def projector_heads(feat, heads, normalize=True):
    """``heads``: list of (w1, b1, w2, b2) of identical shapes (at most 4) -> list of z [N, out], one per head"""
    flat = [t for h in heads for t in h]
    return list(_ProjectorHeadsFn.apply(feat, normalize, *flat))


class _FlipBatchFn(torch.autograd.Function):
    """per-sample H / W flips of an [N,C,H,W] batch (spcl_flip_batch); a flip is its own inverse, so is its gradient"""

    @staticmethod
    def forward(ctx, x, flags):
        xc = x.detach().contiguous()
        out = torch.empty_like(xc)
        N, C, H, W = xc.shape
        _n.call("spcl_flip_batch", _n.ptr(xc), _n.ptr(out), xc.element_size(), N, C, H, W, _n.ptr(flags), _n.stream())
        ctx.flags = flags
        return out

    @staticmethod
    def backward(ctx, g):
        gc = g.contiguous()
        out = torch.empty_like(gc)
        N, C, H, W = gc.shape
        _n.call("spcl_flip_batch", _n.ptr(gc), _n.ptr(out), gc.element_size(), N, C, H, W, _n.ptr(ctx.flags), _n.stream())
        return out, None


def flip_batch(x, flags):
    """differentiable form of TensorRandomFlip.apply_batch: ``flags`` uint8 [N] (bit 0 flip H, bit 1 flip W)"""
    return _FlipBatchFn.apply(x, flags)


class _AdaptivePoolFn(torch.autograd.Function):
    """nn.AdaptiveAvgPool2d / nn.AdaptiveMaxPool2d((oh, ow)) on NHWC storage -> f32 logical [N,C,oh,ow]"""

    @staticmethod
    def forward(ctx, x, out_hw, mode):
        _n.require_gpu(x)
        xs, cs = as_nhwc(x.detach())
        N, H, W, _ = xs.shape
        C = x.shape[1]
        oh, ow = out_hw
        out = torch.empty(N, oh, ow, C, dtype=torch.float32, device=x.device)
        arg = torch.empty(N, oh, ow, C, dtype=torch.int32, device=x.device) if mode == 1 else None
        _n.call("spcl_adaptive_pool2d_forward", _n.ptr(xs), _n.dtype_code(xs.dtype), N, H, W, C, cs, oh, ow, mode,
                _n.ptr(out), _n.ptr(arg), _n.stream())
        ctx.arg = arg
        ctx.meta = (N, H, W, C, cs, oh, ow, mode, xs.dtype, x.dtype)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dout):
        N, H, W, C, cs, oh, ow, mode, sdt, xdt = ctx.meta
        do = _class_map_storage(dout)
        dx = torch.empty(N, H, W, cs, dtype=sdt, device=dout.device)
        _n.call("spcl_adaptive_pool2d_backward", _n.ptr(do), _n.ptr(ctx.arg), _n.dtype_code(sdt), N, H, W, C, cs, oh, ow,
                mode, _n.ptr(dx), _n.stream())
        g = nhwc_to_logical(dx, C)
        return (g if g.dtype == xdt else g.to(xdt)), None, None


def adaptive_pool2d(x, out_hw, mode="avg"):
    return _AdaptivePoolFn.apply(x, (int(out_hw[0]), int(out_hw[1])), 1 if mode == "max" else 0)


class _L2NormChannelsFn(torch.autograd.Function):
    """F.normalize(x, p=2, dim=1) of a logical [N,C,H,W] f32 map with NHWC storage (rows = pixels)"""

    @staticmethod
    def forward(ctx, x):
        _n.require_gpu(x)
        xs = _class_map_storage(x.detach())
        N, H, W, C = xs.shape
        z = torch.empty_like(xs)
        _n.call("spcl_l2norm_rows_forward", _n.ptr(xs), N * H * W, C, _n.ptr(z), _n.stream())
        ctx.save_for_backward(xs)
        return z.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dz):
        (xs,) = ctx.saved_tensors
        N, H, W, C = xs.shape
        dzs = _class_map_storage(dz)
        dx = torch.empty_like(xs)
        _n.call("spcl_l2norm_rows_backward", _n.ptr(xs), _n.ptr(dzs), N * H * W, C, _n.ptr(dx), _n.stream())
        return dx.permute(0, 3, 1, 2)


def l2norm_channels(x):
    return _L2NormChannelsFn.apply(x)


class _L2NormRowsFn(torch.autograd.Function):
    """F.normalize(x, p=2, dim=1) of [rows, d] f32 rows (projectors/nn.py:29-36)"""

    @staticmethod
    def forward(ctx, x):
        _n.require_gpu(x)
        xc = x.detach().contiguous().float()
        z = torch.empty_like(xc)
        _n.call("spcl_l2norm_rows_forward", _n.ptr(xc), xc.shape[0], xc.shape[1], _n.ptr(z), _n.stream())
        ctx.save_for_backward(xc)
        ctx.in_dtype = x.dtype
        return z

    @staticmethod
    def backward(ctx, dz):
        (xc,) = ctx.saved_tensors
        dx = torch.empty_like(xc)
        _n.call("spcl_l2norm_rows_backward", _n.ptr(xc), _n.ptr(dz.detach().contiguous().float()), xc.shape[0], xc.shape[1],
                _n.ptr(dx), _n.stream())
        return dx.to(ctx.in_dtype)


def l2norm_rows(x):
    assert x.dim() == 2, x.shape
    return _L2NormRowsFn.apply(x)


class _PixelMlpFn(torch.autograd.Function):
    """The dense projector's 1x1-conv MLP on every pixel as matrix products over the N*H*W pixel rows (csrc/rows_mlp.hip:
    exact-f32 MFMA; forward, input gradient, weight gradient with a fixed-order fold).  The feature map is read in place
    (channels-last rows of pitch ``cs``), the hidden pre-activation is the only tensor kept for backward."""

    @staticmethod
    def forward(ctx, feat, w1, b1, w2, b2):
        _n.require_gpu(feat, w1, b1, w2, b2)
        x, cs = as_nhwc(feat.detach())
        N, H, W, _ = x.shape
        C, M, dev = feat.shape[1], N * H * W, feat.device
        mlp = w2 is not None
        w1c, b1c = w1.detach().reshape(w1.shape[0], -1).contiguous().float(), b1.detach().contiguous().float()
        first = torch.empty(M, w1c.shape[0], dtype=torch.float32, device=dev)  # pre-activation (mlp) or the output (linear)
        _n.call("spcl_rows_linear_forward", _n.ptr(x), _n.dtype_code(x.dtype), cs, 0, _n.ptr(w1c), _n.ptr(b1c), M, C,
                w1c.shape[0], _n.ptr(first), _n.stream())
        w2c = None
        out = first
        if mlp:
            w2c, b2c = w2.detach().reshape(w2.shape[0], -1).contiguous().float(), b2.detach().contiguous().float()
            out = torch.empty(M, w2c.shape[0], dtype=torch.float32, device=dev)
            _n.call("spcl_rows_linear_forward", _n.ptr(first), _n.dtype_code(torch.float32), w1c.shape[0], 1, _n.ptr(w2c),
                    _n.ptr(b2c), M, w1c.shape[0], w2c.shape[0], _n.ptr(out), _n.stream())
        ctx.save_for_backward(x, w1c, w2c, first if mlp else None)
        ctx.meta = (N, H, W, C, cs, x.dtype, feat.dtype)
        ctx.params = (w1, b1, w2, b2)
        return out.view(N, H, W, out.shape[1]).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dout):
        x, w1c, w2c, pre = ctx.saved_tensors
        N, H, W, C, cs, xdt, fdt = ctx.meta
        M, dev = N * H * W, dout.device
        mlp = w2c is not None
        g = _class_map_storage(dout.detach())  # [N, H, W, O] f32, contiguous
        g = g.view(M, g.shape[3])
        ng = ctx.needs_input_grad
        sk = tuple(take_grad_sink(p, ng[i + 1]) for i, p in enumerate(ctx.params))
        hid = w1c.shape[0]
        dw1 = _grad_buffer(sk[0], (hid, C), dev)
        db1 = _grad_buffer(sk[1], (hid,), dev)
        dw2 = db2 = None

        def wgrad(gm, xin, xdtype, ldx, leaky, n_out, k_in, dw, db):
            ws = torch.empty(_n.call("spcl_rows_linear_backward_weight_workspace_bytes", M, n_out, k_in) // 4 + 1,
                             dtype=torch.float32, device=dev)
            _n.call("spcl_rows_linear_backward_weight", _n.ptr(gm), _n.dtype_code(gm.dtype), _n.ptr(xin), _n.dtype_code(xdtype), ldx,
                    int(leaky), M, n_out, k_in, _n.ptr(ws), ws.numel() * 4, _n.ptr(dw), _n.ptr(db), _n.stream())

        if mlp:
            O = w2c.shape[0]
            dw2 = _grad_buffer(sk[2], (O, hid), dev)
            db2 = _grad_buffer(sk[3], (O,), dev)
            wgrad(g, pre, torch.float32, hid, True, O, hid, dw2, db2)
            dpre = torch.empty(M, hid, dtype=torch.float32, device=dev)
            _n.call("spcl_rows_linear_backward_input", _n.ptr(g), _n.dtype_code(g.dtype), _n.ptr(w2c), _n.ptr(pre), M, O, hid,
                    _n.ptr(dpre), _n.dtype_code(torch.float32), hid, _n.stream())
            g = dpre
        wgrad(g, x, xdt, cs, False, hid, C, dw1, db1)
        gfeat = None
        if ng[0]:
            dfeat = (torch.zeros if cs != C else torch.empty)(N, H, W, cs, dtype=xdt, device=dev)
            _n.call("spcl_rows_linear_backward_input", _n.ptr(g), _n.dtype_code(g.dtype), _n.ptr(w1c), None, M, hid, C, _n.ptr(dfeat),
                    _n.dtype_code(xdt), cs, _n.stream())
            gfeat = nhwc_to_logical(dfeat, C)
            if gfeat.dtype != fdt:
                gfeat = gfeat.to(fdt)
        return gfeat, dw1.view(ctx.params[0].shape), db1, None if dw2 is None else dw2.view(ctx.params[2].shape), db2


class _PixelMlpPooledFn(torch.autograd.Function):
    """``adaptive_avg_pool(conv2(leaky(conv1(x))))`` of the dense projector computed as ``conv2(adaptive_avg_pool(leaky(conv1(x))))``:
    the second 1x1 convolution is linear (bias included: an average of a constant is the constant), so it commutes with the
    average pooling and runs on the N * oh * ow pooled rows instead of the N * H * W pixels -- 31x fewer rows for a 56^2 map
    pooled to 10^2, 125x for 112^2; the products of the wide layer (hid x out) and its full-size output disappear.  Same
    function; the sums are associated differently (differences at the 1e-7 level, tests/test_gpu_round2_heads.py)."""

    @staticmethod
    def forward(ctx, feat, w1, b1, w2, b2, out_hw):
        _n.require_gpu(feat, w1, b1, w2, b2)
        x, cs = as_nhwc(feat.detach())
        N, H, W, _ = x.shape
        C, M, dev = feat.shape[1], N * H * W, feat.device
        oh, ow = out_hw
        f32 = _n.dtype_code(torch.float32)
        w1c, b1c = w1.detach().reshape(w1.shape[0], -1).contiguous().float(), b1.detach().contiguous().float()
        w2c, b2c = w2.detach().reshape(w2.shape[0], -1).contiguous().float(), b2.detach().contiguous().float()
        hid, O = w1c.shape[0], w2c.shape[0]
        # LeakyReLU(conv1(x)): the one full-size tensor kept -- in the feature map's own dtype (3.1 GB as f32 for 60 maps of
        # 224^2: the head's launches are streams of this tensor)
        h = torch.empty(M, hid, dtype=x.dtype, device=dev)
        _n.call("spcl_rows_linear_forward_act", _n.ptr(x), _n.dtype_code(x.dtype), cs, _n.ptr(w1c), _n.ptr(b1c), M, C, hid,
                _n.ptr(h), _n.dtype_code(h.dtype), _n.stream())
        hp = torch.empty(N * oh * ow, hid, dtype=torch.float32, device=dev)
        _n.call("spcl_adaptive_pool2d_forward", _n.ptr(h), _n.dtype_code(h.dtype), N, H, W, hid, hid, oh, ow, 0, _n.ptr(hp), None,
                _n.stream())
        out = torch.empty(N * oh * ow, O, dtype=torch.float32, device=dev)
        _n.call("spcl_rows_linear_forward", _n.ptr(hp), f32, hid, 0, _n.ptr(w2c), _n.ptr(b2c), N * oh * ow, hid, O, _n.ptr(out),
                _n.stream())
        ctx.save_for_backward(x, h, hp, w1c, w2c)
        ctx.meta = (N, H, W, C, cs, oh, ow, x.dtype, feat.dtype)
        ctx.params = (w1, b1, w2, b2)
        return out.view(N, oh, ow, O).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dout):
        x, h, hp, w1c, w2c = ctx.saved_tensors
        N, H, W, C, cs, oh, ow, xdt, fdt = ctx.meta
        M, Mp, dev = N * H * W, N * oh * ow, dout.device
        hid, O = w1c.shape[0], w2c.shape[0]
        f32 = _n.dtype_code(torch.float32)
        g = _class_map_storage(dout.detach()).view(Mp, O)
        ng = ctx.needs_input_grad
        sk = tuple(take_grad_sink(p, ng[i + 1]) for i, p in enumerate(ctx.params))
        dw1, db1 = _grad_buffer(sk[0], (hid, C), dev), _grad_buffer(sk[1], (hid,), dev)
        dw2, db2 = _grad_buffer(sk[2], (O, hid), dev), _grad_buffer(sk[3], (O,), dev)

        def wgrad(gm, rows, xin, xdtype, ldx, n_out, k_in, dw, db):
            ws = torch.empty(_n.call("spcl_rows_linear_backward_weight_workspace_bytes", rows, n_out, k_in) // 4 + 1,
                             dtype=torch.float32, device=dev)
            _n.call("spcl_rows_linear_backward_weight", _n.ptr(gm), _n.dtype_code(gm.dtype), _n.ptr(xin), _n.dtype_code(xdtype), ldx, 0,
                    rows, n_out, k_in, _n.ptr(ws), ws.numel() * 4, _n.ptr(dw), _n.ptr(db), _n.stream())

        wgrad(g, Mp, hp, torch.float32, hid, O, hid, dw2, db2)
        dhp = torch.empty(Mp, hid, dtype=torch.float32, device=dev)
        _n.call("spcl_rows_linear_backward_input", _n.ptr(g), f32, _n.ptr(w2c), None, Mp, O, hid, _n.ptr(dhp), f32, hid, _n.stream())
        dpre = torch.empty(M, hid, dtype=h.dtype, device=dev)  # un-pooled, through the LeakyReLU (h carries its sign)
        _n.call("spcl_adaptive_avgpool2d_backward_act", _n.ptr(dhp), _n.ptr(h), _n.dtype_code(h.dtype), N, H, W, hid, oh, ow,
                _n.ptr(dpre), _n.stream())
        wgrad(dpre, M, x, xdt, cs, hid, C, dw1, db1)
        gfeat = None
        if ng[0]:
            dfeat = (torch.zeros if cs != C else torch.empty)(N, H, W, cs, dtype=xdt, device=dev)
            _n.call("spcl_rows_linear_backward_input", _n.ptr(dpre), _n.dtype_code(dpre.dtype), _n.ptr(w1c), None, M, hid, C,
                    _n.ptr(dfeat), _n.dtype_code(xdt), cs, _n.stream())
            gfeat = nhwc_to_logical(dfeat, C)
            if gfeat.dtype != fdt:
                gfeat = gfeat.to(fdt)
        return gfeat, dw1.view(ctx.params[0].shape), db1, dw2.view(ctx.params[2].shape), db2, None


def pixelwise_mlp_pooled_supported(feat, w1, w2) -> bool:
    return bool(feat.is_cuda and w2 is not None and feat.shape[1] % 4 == 0 and w1.shape[0] % 4 == 0 and w2.shape[0] % 4 == 0)


def pixelwise_mlp_pooled(feat, w1, b1, w2, b2, out_hw):
    """``adaptive_avg_pool2d(pixelwise_mlp(feat, ...), out_hw)`` with the second layer applied AFTER the pooling (see
    ``_PixelMlpPooledFn``) -> f32 logical [N, O, oh, ow]"""
    return _PixelMlpPooledFn.apply(feat, w1, b1, w2, b2, tuple(int(v) for v in out_hw))


def pixelwise_mlp(feat, w1, b1, w2=None, b2=None):
    """1x1-conv MLP of ``get_contrastive_dense_projector`` (projectors/heads.py:28-39) on a logical [N,C,H,W] map:
    every pixel is a row (no pooling, no normalisation) -> f32 logical [N,O,H,W].
    ``w1`` [hid,C,1,1] (or [O,C,1,1] for the linear head), ``w2`` [O,hid,1,1].  Matrix products over the pixel rows
    (``_PixelMlpFn``) when the channel counts are multiples of 4 -- always, in the UNet --; else the rows go through the
    global projector's kernels (correct, and slow beyond a few thousand rows)."""
    dims = [feat.shape[1], w1.shape[0]] + ([w2.shape[0]] if w2 is not None else [])
    if feat.is_cuda and all(d % 4 == 0 for d in dims):
        return _PixelMlpFn.apply(feat, w1, b1, w2, b2)
    xs, cs = as_nhwc(feat)  # differentiable view ops only
    N, H, W, _ = xs.shape
    C = feat.shape[1]
    rows = torch.as_strided(xs, (N * H * W, C, 1, 1), (cs, 1, cs, cs), xs.storage_offset())
    o = projector(rows, w1.reshape(w1.shape[0], -1), b1, None if w2 is None else w2.reshape(w2.shape[0], -1), b2,
                  normalize=False)
    return o.view(N, H, W, o.shape[1]).permute(0, 3, 1, 2)


# --------------------------------------------------------------------------------------------- encoder blocks
def _ru16(c: int) -> int:
    return (c + 15) // 16 * 16


def to_nhwc_padded(x: torch.Tensor, dtype: torch.dtype):
    """logical NCHW -> contiguous [N,H,W,CS] storage of `dtype`, CS = C rounded up to 16 (pad channels = 0).
    Zero-copy when x already is such a view (what the fused blocks hand to each other)."""
    N, C, H, W = x.shape
    cs = _ru16(C)
    if x.dtype == dtype and cs == C:
        sN, sC, sH, sW = x.stride()
        if sC == 1 and sW == cs and sH == W * cs and sN == H * W * cs:
            return x.permute(0, 2, 3, 1)
    # (a channel count that is not a multiple of 16 is always re-padded: a narrow view cannot prove that the lanes
    # behind it hold zeros -- e.g. a channel slice of a concatenation's gradient)
    if cs == C:
        return x.to(dtype).contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)
    out = torch.zeros(N, H, W, cs, dtype=dtype, device=x.device)
    out[..., :C] = x.permute(0, 2, 3, 1)
    return out


def nhwc_channel_slice(x, dtype):
    """``(view [N,H,W,C], S)`` when the logical NCHW tensor ``x`` is a 16-byte-aligned CHANNEL SLICE of a dense NHWC tensor of
    ``dtype`` with S > C channels per pixel (one half of a decoder concatenation's gradient, ``_VirtualCatFn``) -- what the
    ``_strided`` BatchNorm entry points read in place; else None (the caller re-packs)."""
    if x is None or x.dim() != 4 or x.dtype != dtype:
        return None
    N, C, H, W = x.shape
    sN, sC, sH, sW = x.stride()
    if sC != 1 or sW <= C or sW % 8 or C % 16 or sH != W * sW or sN != H * W * sW:
        return None
    if (x.storage_offset() * x.element_size()) % 16:
        return None
    return x.permute(0, 2, 3, 1), sW


def _bnrelu_fwd(y, dt_code, N, H, W, cs, scale, shift, act, pool):
    """BN-apply + ReLU (+ 2x2 max-pool) writer; ``act`` may be a channel slice of a wider NHWC tensor (pixel stride > cs)"""
    if act is not None and act.stride(2) != cs:
        _n.call("spcl_bnrelu_pool_forward_strided", _n.ptr(y), dt_code, N, H, W, cs, _n.ptr(scale), _n.ptr(shift),
                _n.ptr(act), int(act.stride(2)), _n.ptr(pool), _n.stream())
    else:
        _n.call("spcl_bnrelu_pool_forward", _n.ptr(y), dt_code, N, H, W, cs, _n.ptr(scale), _n.ptr(shift), _n.ptr(act),
                _n.ptr(pool), _n.stream())


def _act_buffer(cfg, N, H, W, cs, dtype, dev):
    """the block's activation output: the caller's destination (``cfg.act_dst``: its half of a concatenation buffer) when it
    fits, else a fresh dense tensor"""
    dst = getattr(cfg, "act_dst", None)
    if (dst is not None and tuple(dst.shape) == (N, H, W, cs) and dst.dtype == dtype and dst.device == dev
            and dst.stride(3) == 1 and dst.stride(2) % 8 == 0 and dst.stride(1) == W * dst.stride(2)
            and dst.stride(0) == H * W * dst.stride(2) and (dst.storage_offset() * dst.element_size()) % 16 == 0):
        return dst
    return torch.empty(N, H, W, cs, dtype=dtype, device=dev)


class PoolLink:
    """What the NEXT block's backward needs to leave the BatchNorm-backward partial sums of THIS block's second conv in the
    epilogue of its own input-gradient kernel (spcl_conv3x3_dgrad_poolstats): this block's raw second-conv output and BN
    coefficients.  The next block fills ``rows`` / ``dx_ptr``; this block's backward uses them if the pooled gradient it
    receives IS that kernel's output (same storage: nothing else contributed to it), else runs its own reduction pass."""
    __slots__ = ("yb", "stb", "N", "H", "W", "cout_s", "rows", "dx_ptr", "holder", "version", "acc")

    def __init__(self, yb, stb, N, H, W, cout_s, acc=None):
        self.yb, self.stb, self.N, self.H, self.W, self.cout_s = yb, stb, N, H, W, cout_s
        # ``acc``: a zeroed fixed-point accumulator block (bn_acc_block) the next block's dgrad may ADD the sums to instead of
        # writing per-tile rows; ``rows`` is then ``ACC_ROWS`` and the apply pass derives its coefficients from the block
        self.acc = acc
        # ``holder``: the gradient tensor ``dx_ptr`` is the address of, kept alive until the consumer has compared -- a freed
        # tensor's address is the first thing the allocator hands out again (to the re-layout copy of a foreign gradient, say)
        self.rows, self.dx_ptr, self.holder, self.version = None, 0, None, 0

    def fresh(self, ptr):
        """is the gradient at ``ptr`` exactly the tensor the producer left (same storage, never accumulated into in place)?"""
        return self.rows is not None and ptr == self.dx_ptr and self.holder is not None and self.holder._version == self.version


class ActLink:
    """A block whose activation has ONE consumer that applies the block's last BatchNorm + ReLU itself (``BlockCfg.lazy_act``:
    the 1x1 head, ``conv1x1_bn``): the raw second-conv output and the BN coefficients the consumer needs, and -- filled by
    the consumer's backward -- the BatchNorm-backward partial sums it left next to the activation gradient."""
    __slots__ = ("yb", "stb", "N", "H", "W", "C", "cs", "rows", "dx_ptr", "holder", "version")

    def __init__(self, yb, stb, N, H, W, C, cs):
        self.yb, self.stb, self.N, self.H, self.W, self.C, self.cs = yb, stb, N, H, W, C, cs
        self.rows, self.dx_ptr, self.holder, self.version = None, 0, None, 0  # (holder / version: see PoolLink)

    fresh = PoolLink.fresh


class UpLink:
    """Between a block whose activation has ONE consumer -- an up-convolution reading it at half resolution
    (``BlockCfg.up_in``) -- and that consumer: the consumer's backward leaves the FINE input gradient here instead of summing
    its 2 x 2 windows, and returns a shared half-resolution tensor of ZEROS whose address the block recognises; the block's
    BatchNorm-backward reduction pass forms the sums on its way (spcl_bnrelu_backward_up2).  A gradient that arrives at the
    block with any other address means somebody else contributed to it (zeros + their gradient): the block then adds the
    2 x 2 sums itself with the ordinary kernels."""
    __slots__ = ("d_up", "dx_ptr", "holder", "version")

    def __init__(self):
        self.d_up, self.dx_ptr, self.holder, self.version = None, 0, None, 0


class BlockCfg:
    """Static configuration of one fused Conv-BN-ReLU(-Conv-BN-ReLU)(-MaxPool) block call."""
    __slots__ = ("dtype", "training", "momentum", "eps", "track", "need_act", "need_pool", "image_input", "buffers",
                 "link_in", "link_out", "act_dst", "up2", "lazy_act", "link_act", "x2_link", "bn_link", "x2_bn", "up_in", "up_link",
                 "gap", "want_gap")

    def __init__(self, dtype, training, momentum, eps, track, need_act, need_pool, image_input, buffers):
        self.dtype, self.training, self.momentum, self.eps, self.track = dtype, training, momentum, eps, track
        self.need_act, self.need_pool, self.image_input, self.buffers = need_act, need_pool, image_input, buffers
        self.link_in = None   # PoolLink of the block whose pooled output is this block's input (set by the caller)
        self.link_out = None  # PoolLink this call offers to the next block (set by the forward)
        self.act_dst = None   # [N,H,W,cout_s] view (pixel stride >= cout_s) the activation is written to: one half of a
                              # decoder concatenation buffer (UNet.forward), so that torch.cat needs no copy
        self.lazy_act = False  # the activation's only consumer applies BN + ReLU itself: return the RAW output, offer link_act
        self.link_act = None
        self.x2_link = None    # ActLink of the producer of ``x2`` when that tensor is its RAW output (conv_block(..., x2=...))
        self.up_in = False     # conv_bn_relu: the input is nn.Upsample(x2) of the HALF-resolution tensor passed in (read in place)
        self.up_link = None    # UpLink shared by that conv_bn_relu call and the block that produced its input
        self.bn_link = None    # conv_bn_relu: its (raw output, BN coefficients) offered to the consumer of its activation ...
        self.x2_bn = None      # ... conv_block(..., x2=that activation): its dgrad leaves that BatchNorm's backward sums there
        self.gap = None       # set by the forward: [N, C] f32 global average of the activation it wrote (small maps), see conv_block
        self.want_gap = False  # the caller has a consumer for it (the block's output is tapped by a forward hook: semi_seg/arch/hook.py)
        self.up2 = False      # the activation's only consumer is nn.Upsample(scale_factor=2): write it 2x2-replicated, return
                              # the [N, C, 2H, 2W] tensor (spcl_bnrelu_up2_forward); backward sums the 2x2 gradients first


def _pack(w, kind, dt_code, dtype):
    co, ci = w.shape[0], w.shape[1]
    n = _n.call("spcl_conv_packed_elems", ci, co, kind, dt_code)
    buf = torch.empty(n, dtype=dtype, device=w.device)
    wc = w.detach().contiguous().float()
    _n.call("spcl_conv_pack_weights", _n.ptr(wc), ci, co, kind, dt_code, _n.ptr(buf), _n.stream())
    return buf


def _pack_both(w, dt_code, dtype):
    """forward and dgrad fragment layouts in one launch -> (packed_fwd, packed_dgrad)."""
    co, ci = w.shape[0], w.shape[1]
    n0 = _n.call("spcl_conv_packed_elems", ci, co, 0, dt_code)
    n1 = _n.call("spcl_conv_packed_elems", ci, co, 1, dt_code)
    buf = torch.empty(n0 + n1, dtype=dtype, device=w.device)
    wc = w.detach().contiguous().float()
    p0, p1 = buf[:n0], buf[n0:]
    _n.call("spcl_conv_pack_weights_both", _n.ptr(wc), ci, co, dt_code, _n.ptr(p0), _n.ptr(p1), _n.stream())
    return p0, p1


def _pack_block(wa, wb, dt_code, dtype, H=0, W=0):
    """both convolutions of a block, forward and dgrad layouts, one launch -> ((a_fwd, a_dgrad), (b_fwd, b_dgrad)).
    ``H, W``: the image size they are used at (the band-GEMM layout is only written where that kernel runs)."""
    sizes = []
    for w in (wa, wb):
        co, ci = w.shape[0], w.shape[1]
        sizes += [_n.call("spcl_conv_packed_elems", ci, co, 0, dt_code), _n.call("spcl_conv_packed_elems", ci, co, 1, dt_code)]
    buf = torch.empty(sum(sizes), dtype=dtype, device=wa.device)
    parts, off = [], 0
    for n_ in sizes:
        parts.append(buf[off:off + n_])
        off += n_
    wac, wbc = wa.detach().contiguous().float(), wb.detach().contiguous().float()
    _n.call("spcl_conv_pack_weights_block_at", _n.ptr(wac), wa.shape[1], wa.shape[0], _n.ptr(parts[0]), _n.ptr(parts[1]),
            _n.ptr(wbc), wb.shape[1], wb.shape[0], _n.ptr(parts[2]), _n.ptr(parts[3]), dt_code, int(H), int(W), _n.stream())
    return (parts[0], parts[1]), (parts[2], parts[3])


# ---- weights of several layers packed ahead of their use, in ONE launch (UNet.forward announces the layers it is about to
# run: one pack launch per forward pass instead of one per block).  Entries are consumed by the block that uses them.
_PREPACKED = {}
_PREPACK = os.environ.get("SPCL_PREPACK", "1") != "0"  # A/B switch: 0 = every block packs right before its convolutions


_PREPACKED_ACORR = {}


def take_prepacked_acorr(image):
    return _PREPACKED_ACORR.pop((image.data_ptr(), image._version), None)


_EVAL_AFFINE = {}


def clear_prepacked():
    """drop what a forward pass announced and did not consume (UNet.forward calls it on every exit)"""
    _PREPACKED.clear()
    _PREPACKED_ACORR.clear()
    _EVAL_AFFINE.clear()


def prepare_eval_affines(bns):
    """``bns``: the BatchNorm2d modules an EVAL-mode forward pass is about to apply: mean / invstd / scale / shift of all of
    them from their running statistics in ONE launch (spcl_bn_eval_affine_multi), handed to ``_bn_stats`` through
    ``_EVAL_AFFINE`` (keyed by the running mean's storage).  One launch per layer was 22 launches of ~5 us in a validation
    batch of the full UNet: a quarter of its GPU time."""
    _EVAL_AFFINE.clear()
    bns = [bn for bn in bns if bn.running_mean is not None and bn.running_mean.is_cuda and bn.weight is not None]
    for i in range(0, len(bns), _n.BN_EVAL_MAX):
        part = bns[i:i + _n.BN_EVAL_MAX]
        dev = part[0].running_mean.device
        sizes = [_ru16(bn.num_features) for bn in part]
        buf = torch.empty(4 * sum(sizes), dtype=torch.float32, device=dev)
        items, keep, off = [], [], 0
        for bn, cs in zip(part, sizes):
            g, b = bn.weight.detach().contiguous().float(), bn.bias.detach().contiguous().float()
            st = buf[off:off + 4 * cs].view(4, cs)
            off += 4 * cs
            items.append(_n.BnEvalItem(g.data_ptr(), b.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                                       st.data_ptr(), bn.num_features, cs, float(bn.eps)))
            keep.append((g, b))
            _EVAL_AFFINE[(bn.running_mean.data_ptr(), cs)] = st
        arr = (_n.BnEvalItem * len(items))(*items)
        _n.call("spcl_bn_eval_affine_multi", arr, len(items), _n.stream())


def prepack_weights(layers, dtype, image=None):
    """``layers``: list of (weight [Cout, Cin, 3, 3], H, W) about to be used at image size H x W with storage ``dtype``:
    forward and dgrad layouts of all of them in one launch (spcl_conv_pack_weights_multi), handed to the blocks through
    ``take_prepacked``.  Anything left over from an earlier, unfinished forward is dropped.
    ``image`` [N, H, W] (or [N, H, W, 1]) f32: the same launch also computes its autocorrelation rows (``_image_autocorr``,
    the image3 path of the first block), handed over through ``take_prepacked_acorr``."""
    _PREPACKED.clear()
    _PREPACKED_ACORR.clear()
    if not _PREPACK or not layers:
        return
    dtc = _n.dtype_code(dtype)
    dev = layers[0][0].device
    sizes = []
    for w, H, W in layers:
        co, ci = w.shape[0], w.shape[1]
        sizes.append((_n.call("spcl_conv_packed_elems", ci, co, 0, dtc), _n.call("spcl_conv_packed_elems", ci, co, 1, dtc)))
    for i in range(0, len(layers), _n.PACK_MULTI_MAX):
        part, psz = layers[i:i + _n.PACK_MULTI_MAX], sizes[i:i + _n.PACK_MULTI_MAX]
        buf = torch.empty(sum(a + b for a, b in psz), dtype=dtype, device=dev)
        items, keep, off = [], [], 0
        for (w, H, W), (n0, n1) in zip(part, psz):
            wc = w.detach().contiguous().float()
            p0, p1 = buf[off:off + n0], buf[off + n0:off + n0 + n1]
            off += n0 + n1
            items.append(_n.PackItem(wc.data_ptr(), p0.data_ptr(), p1.data_ptr(), w.shape[1], w.shape[0],
                                     int(H) if _PACK_AT else 0, int(W) if _PACK_AT else 0))
            keep.append(wc)
            _PREPACKED[(w.data_ptr(), w._version, dtc, int(H), int(W))] = (p0, p1)  # (_version: an in-place update since)
        arr = (_n.PackItem * len(items))(*items)
        if image is not None and i == 0 and _acorr_in_conv_rows(dtc, *[int(v) for v in image.shape[:3]]) > 0:
            image = None  # (the image convolution leaves the autocorrelation rows itself: _conv_image_acorr)
        # (the pass's BatchNorm accumulator arena is zeroed by this launch -- the first of the pass -- instead of a fill launch)
        zero = _acc_arena_take_dirty(dev) if i == 0 else None
        zbytes = ctypes.c_size_t(zero.numel() * 8 if zero is not None else 0)
        if image is not None and i == 0:
            N, H, W = int(image.shape[0]), int(image.shape[1]), int(image.shape[2])
            acorr = torch.empty(_n.call("spcl_image_autocorr_rows", N, H, W), 64, dtype=torch.float32, device=dev)
            _n.call("spcl_conv_pack_weights_multi_zero", arr, len(items), dtc, _n.ptr(image), N, H, W, _n.ptr(acorr),
                    _n.ptr(zero), zbytes, _n.stream())
            _PREPACKED_ACORR[(image.data_ptr(), image._version)] = acorr
        else:
            _n.call("spcl_conv_pack_weights_multi_zero", arr, len(items), dtc, None, 0, 0, 0, None, _n.ptr(zero), zbytes,
                    _n.stream())


def _acorr_in_conv_rows(dt_code, N, H, W, cin=1, cout_s=16):
    """rows ([rows][64] f32) the image block's first convolution leaves as the image's autocorrelation partials when it
    runs as spcl_conv3x3_forward_image_acorr; 0: that form does not take the shape (the rows come from their own pass)"""
    return _n.call("spcl_conv3x3_forward_image_acorr_rows", dt_code, N, H, W, cin, cout_s)


def _conv_image_acorr(x_store, dt_code, dtype, N, H, W, cin_s, cout_s, wp, want_stats, rows):
    """the one-channel image convolution + the image's autocorrelation rows in ONE launch
    (spcl_conv3x3_forward_image_acorr) -> (y, stats, acorr [rows][64])"""
    dev = x_store.device
    y = torch.empty(N, H, W, cout_s, dtype=dtype, device=dev)
    stats = None
    if want_stats:
        nt = _n.call("spcl_conv_stat_rows", dt_code, N, H, W, 16, cout_s)
        stats = torch.empty(_n.call("spcl_bn_stats_elems", nt, cout_s), dtype=torch.float32, device=dev)
        stats.ntiles = nt
    acorr = torch.empty(rows, 64, dtype=torch.float32, device=dev)
    _n.call("spcl_conv3x3_forward_image_acorr", _n.ptr(x_store), dt_code, N, H, W, cin_s, cout_s, _n.ptr(wp), _n.ptr(y),
            _n.ptr(stats), _n.ptr(acorr), _n.stream())
    return y, stats, acorr


def take_prepacked(w, dtc, H, W):
    return _PREPACKED.pop((w.data_ptr(), w._version, dtc, int(H), int(W)), None)


def _conv(x_store, dt_code, dtype, N, H, W, cin_s, cin_k, cout_s, wp, in_mode, scale, shift, want_stats):
    dev = x_store.device
    y = torch.empty(N, H, W, cout_s, dtype=dtype, device=dev)
    stats = None
    if want_stats:
        # rows [tile][3][cout_s] of Chan partials (+ the finalize kernel's scratch rows), see include/spcl_hip.h
        nt = _n.call("spcl_conv_stat_rows", dt_code, N, H, W, cin_k, cout_s)
        stats = torch.empty(_n.call("spcl_bn_stats_elems", nt, cout_s), dtype=torch.float32, device=dev)
        stats.ntiles = nt
    _n.call("spcl_conv3x3_forward", _n.ptr(x_store), dt_code, N, H, W, cin_s, cin_k, cout_s, _n.ptr(wp), in_mode,
            _n.ptr(scale), _n.ptr(shift), _n.ptr(y), _n.ptr(stats), _n.stream())
    return y, stats


def _conv_cat(xa, xb, dt_code, dtype, N, H, W, chalf, cout_s, wp, want_stats, xb_scale=None, xb_shift=None):
    """the convolution of ``torch.cat((xa, xb), channel)`` read from the two tensors (spcl_conv3x3_forward_cat);
    ``xb_scale`` / ``xb_shift``: ``xb`` is a raw convolution output, the concatenation holds relu(scale xb + shift)"""
    dev = xa.device
    y = torch.empty(N, H, W, cout_s, dtype=dtype, device=dev)
    stats = None
    if want_stats:
        nt = _n.call("spcl_conv_stat_rows", dt_code, N, H, W, 2 * chalf, cout_s)
        stats = torch.empty(_n.call("spcl_bn_stats_elems", nt, cout_s), dtype=torch.float32, device=dev)
        stats.ntiles = nt
    _n.call("spcl_conv3x3_forward_cat", _n.ptr(xa), _n.ptr(xb), dt_code, N, H, W, chalf, cout_s, _n.ptr(wp),
            _n.ptr(xb_scale), _n.ptr(xb_shift), _n.ptr(y), _n.ptr(stats), _n.stream())
    return y, stats


# ---- BatchNorm sums as fixed-point accumulator blocks (csrc/bn_acc.hpp): the producing kernel's epilogue ADDS its tile's sums
# (exact integer atomics: order-free, bit-for-bit deterministic), the next launch derives the coefficients in its prologue -- no
# finalize launch between them.  A block must be ZERO before its producer runs: all blocks of a forward pass (and of the
# backward that follows it) are slices of ONE zero-filled arena, i.e. one fill launch per step.
_BN_ACC = os.environ.get("SPCL_BN_ACC", "1") != "0"  # A/B switch: 0 keeps the per-tile rows + finalize launches everywhere
_ACC_ARENA = {"buf": None, "off": 0, "used": 0, "need": 0, "dirty": False}


def bn_acc_arena_begin(device):
    """a new forward pass starts (UNet.forward): an arena of what the previous pass used, zeroed by the pass's weight-pack
    launch (``prepack_weights``) or, failing that, by a fill of its own at the first block taken; blocks beyond it (the first
    pass, a changed shape) come from on-demand zero-filled chunks"""
    a = _ACC_ARENA
    a["need"], a["used"], a["off"] = a["used"], 0, 0
    a["buf"] = torch.empty(a["need"], dtype=torch.int64, device=device) if (a["need"] > 0 and _BN_ACC) else None
    a["dirty"] = a["buf"] is not None


def _acc_arena_take_dirty(device):
    """the arena still to be zeroed, for a launch that can do it on its way (-> tensor or None); the caller MUST zero it"""
    a = _ACC_ARENA
    if a["dirty"] and a["buf"] is not None and a["buf"].device == device:
        a["dirty"] = False
        return a["buf"]
    return None


def bn_acc_block(cs, device):
    """a zeroed accumulator block for a BatchNorm of ``cs`` (padded) channels -> int64 tensor [spcl_bn_acc_elems(cs)]"""
    words = _n.call("spcl_bn_acc_elems", int(cs))
    a = _ACC_ARENA
    buf = a["buf"]
    if buf is not None and a["dirty"]:  # nobody zeroed it on the way (no weight-pack launch in this pass)
        buf.zero_()
        a["dirty"] = False
    if buf is None or buf.device != device or a["off"] + words > buf.numel():
        buf = a["buf"] = torch.zeros(max(words, 1 << 15), dtype=torch.int64, device=device)
        a["off"] = 0
    blk = buf[a["off"]:a["off"] + words]
    a["off"] += words
    a["used"] += words
    return blk


def _bn_acc_desc(acc, cfg: "BlockCfg", C, cs, gamma, beta, which, count, st, keep):
    """the ``spcl_bn_acc`` of one BatchNorm (native.BnAcc); ``keep`` collects the tensors its pointers borrow"""
    g, b = gamma.detach().contiguous().float(), beta.detach().contiguous().float()
    rm, rv, nbt = cfg.buffers[which]
    upd = cfg.track[which]
    keep.extend([g, b])
    return _n.BnAcc(acc.data_ptr(), g.data_ptr(), b.data_ptr(), rm.data_ptr() if upd else None,
                    rv.data_ptr() if upd else None, nbt.data_ptr() if (upd and nbt is not None) else None, st.data_ptr(),
                    float(cfg.momentum), float(cfg.eps), float(count), int(C), int(cs))


def _conv_acc(x_store, dt_code, dtype, N, H, W, cin_k, cout_s, wp, in_bn, in_scale, in_shift, stats_acc, want_rows):
    """spcl_conv3x3_forward_acc: the input's BatchNorm + ReLU from a block (``in_bn``: native.BnAcc) or from scale / shift
    arrays or none; the output's statistics into ``stats_acc`` or, ``want_rows``, per-tile rows -> (y, rows or None)"""
    dev = x_store.device
    y = torch.empty(N, H, W, cout_s, dtype=dtype, device=dev)
    rows = None
    if want_rows and stats_acc is None:
        nt = _n.call("spcl_conv_stat_rows", dt_code, N, H, W, cin_k, cout_s)
        rows = torch.empty(_n.call("spcl_bn_stats_elems", nt, cout_s), dtype=torch.float32, device=dev)
        rows.ntiles = nt
    _n.call("spcl_conv3x3_forward_acc", _n.ptr(x_store), dt_code, N, H, W, cin_k, cout_s, _n.ptr(wp),
            ctypes.byref(in_bn) if in_bn is not None else None, _n.ptr(in_scale), _n.ptr(in_shift), _n.ptr(y),
            _n.ptr(stats_acc), _n.ptr(rows), _n.stream())
    return y, rows


def _bn_stats(stats, cfg: BlockCfg, C, cs, gamma, beta, which, dev):
    """-> tensor [4, cs]: mean, invstd, scale, shift."""
    st = torch.empty(4, cs, dtype=torch.float32, device=dev)
    g, b = gamma.detach().contiguous().float(), beta.detach().contiguous().float()
    rm, rv, nbt = cfg.buffers[which]
    if cfg.training:
        upd = cfg.track[which]
        _n.call("spcl_bn_finalize", _n.ptr(stats), stats.ntiles, C, cs, _n.ptr(g), _n.ptr(b), c_float(cfg.momentum),
                c_float(cfg.eps), _n.ptr(rm if upd else None), _n.ptr(rv if upd else None),
                _n.ptr(nbt if upd else None), _n.ptr(st[0]), _n.ptr(st[1]), _n.ptr(st[2]), _n.ptr(st[3]), _n.stream())
    else:
        pre = _EVAL_AFFINE.pop((rm.data_ptr(), cs), None) if rm.dtype == torch.float32 else None
        if pre is not None:  # computed with the pass's other layers in one launch (prepare_eval_affines)
            return pre
        _n.call("spcl_bn_eval_affine", C, cs, _n.ptr(g), _n.ptr(b), _n.ptr(rm), _n.ptr(rv), c_float(cfg.eps),
                _n.ptr(st[0]), _n.ptr(st[1]), _n.ptr(st[2]), _n.ptr(st[3]), _n.stream())
    return st


class DeferredWgrads:
    """Weight gradients of the wide (channel counts multiples of 64) bf16 layers, queued during backward and computed by
    ONE batched launch (csrc/wgrad_gemm.hip: the split-K partials are then written once per CU for all layers together
    instead of once per CU and layer).

    Only gradients that go STRAIGHT INTO A FLAT GRADIENT BUCKET are deferred (the parameter's armed sink, see
    ``take_grad_sink``), and autograd never sees them: the backward returns ``None`` for that parameter and the batched
    launch writes the bucket slice when the queue is flushed.  ``ddp.GradBucket.arm_sinks`` opens the queue, ``gather``
    flushes it before anything reads the bucket, adds what autograd accumulated from OTHER uses of the same parameter in
    that step (they do not get the sink) and points ``param.grad`` at the slice."""

    def __init__(self):
        self.items, self.keep, self.targets = [], [], set()
        self.tails = []

    def add(self, x_store, dy, scale, shift, sink, N, H, W, cin, cin_s, cout, cout_s, in_mode, x2=None, x_up2=False):
        it = _n.WgradItem(x_store.data_ptr(), dy.data_ptr(), scale.data_ptr() if scale is not None else None,
                          shift.data_ptr() if shift is not None else None, sink.data_ptr(), N, H, W, cin, cin_s, cout,
                          cout_s, in_mode, x2.data_ptr() if x2 is not None else None, 1 if x_up2 else 0)
        self.items.append(it)
        self.keep.append((x_store, dy, scale, shift, x2))  # operands stay alive until the launch
        self.targets.add(sink.data_ptr())
        if len(self.items) == _QUEUE_MAX[0]:
            self.flush()

    def capture_tail(self, sink, keep, launch):
        """Run ``launch()`` -- one of the narrow layers' weight-gradient producers writing to ``sink`` -- with a tail
        capture armed (``spcl_wgrad_tail_capture``): its few-microsecond final reduction launch is skipped and rides in
        the batched launch's reduction kernel at the next flush.  ``keep``: the buffers the pending sum reads.  Returns
        True when the producer left a tail (the gradient then reaches ``sink`` at the flush, like a deferred item)."""
        tail = _n.WgradTail()
        _n.call("spcl_wgrad_tail_capture", ctypes.byref(tail))
        try:
            launch()
        finally:
            _n.call("spcl_wgrad_tail_capture", None)
        if tail.kind < 0:
            return False
        self.tails.append(tail)
        self.keep.append(keep)
        self.targets.add(sink.data_ptr())
        if len(self.tails) == _QUEUE_MAX[1]:
            self.flush()
        return True

    def flush(self):
        if not self.items and not self.tails:
            return
        wgrad_batched(self.items, accumulate=False, device=self.keep[0][1].device, tails=self.tails)
        self.items, self.keep, self.tails = [], [], []


# items / tails a queue holds before it flushes by itself (A/B switch SPCL_WGRAD_QUEUE_MAX; the library's limits by default:
# the whole UNet's eleven wide layers and ten narrow ones then leave in ONE batched launch at the gather)
_QUEUE_MAX = (min(_n.WGRAD_BATCH_MAX, int(os.environ.get("SPCL_WGRAD_QUEUE_MAX", _n.WGRAD_BATCH_MAX))),
              min(_n.WGRAD_TAILS_MAX, int(os.environ.get("SPCL_WGRAD_QUEUE_MAX", _n.WGRAD_TAILS_MAX))))
_CONV_UP2 = os.environ.get("SPCL_CONV_UP2", "1") != "0"  # A/B switch: 0 writes the x2-upsampled activations (BlockCfg.up2)
_UP2_BWD_FUSED = os.environ.get("SPCL_UP2_BWD_FUSED", "1") != "0"  # A/B switch: 0 sums the 2x2 gradients in a launch of its own
_CONV_SPLIT = os.environ.get("SPCL_CONV_SPLIT", "1") != "0"  # A/B switch: 0 leaves that level's gradient as one interleaved tensor
_CONV_CAT = os.environ.get("SPCL_CONV_CAT", "1") != "0"  # A/B switch: 0 materialises the 16-channel decoder concatenation
_PACK_AT = os.environ.get("SPCL_PACK_AT", "1") != "0"  # A/B switch: 0 packs the band-GEMM layout whether or not it is used
_ACC_FILL = os.environ.get("SPCL_ACC_FILL", "1") != "0"  # A/B switch: 0 keeps the finalize launch of the generic BatchNorm backward
_GAP = os.environ.get("SPCL_GAP", "1") != "0"  # A/B switch: 0 leaves the global average to the projector's own pooling launch
_TAILS = os.environ.get("SPCL_WGRAD_TAILS", "1") != "0"  # A/B switch: 0 keeps every layer's own final reduction launch


def sink_queue(sink):
    """the deferred weight-gradient queue of the bucket that armed ``sink`` (``ddp.GradBucket.arm_sinks`` hangs it on the
    bucket's views), or None: every bucket owns its queue, so two buckets armed in one step never see each other's items
    and a step that was aborted between arming and gathering leaves nothing behind for the next one"""
    return getattr(sink, "_spcl_queue", None) if sink is not None else None


def wgrad_batched(items, accumulate, device, tails=()):
    """one launch for ``items`` (list of native.WgradItem, at most native.WGRAD_BATCH_MAX) and the captured ``tails``
    (native.WgradTail, at most native.WGRAD_TAILS_MAX)"""
    arr, ws = None, None
    if items:
        arr = (_n.WgradItem * len(items))(*items)
        nbytes = _n.call("spcl_conv_wgrad_batched_workspace_bytes", arr, len(items))
        if nbytes == 0:
            raise RuntimeError("wgrad_batched: unsupported item (bf16 NHWC, channel counts multiples of 64, in_mode 0/1)")
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
    tarr = (_n.WgradTail * len(tails))(*tails) if tails else None
    _n.call("spcl_conv3x3_wgrad_batched_tails", arr, len(items), tarr, len(tails), int(bool(accumulate)), _n.ptr(ws),
            _n.stream())


def _wgrad(x_store, dy, dt_code, N, H, W, cin, cin_s, cin_k, cout, cout_s, in_mode, scale, shift, sink=None):
    dev = dy.device
    queue = sink_queue(sink)
    if (queue is not None and cin_s == cin_k
            and _n.call("spcl_conv_wgrad_batched_supported", dt_code, cin, cin_s, cout, cout_s, in_mode)):
        queue.add(x_store, dy, scale, shift, sink, N, H, W, cin, cin_s, cout, cout_s, in_mode)
        return None  # written into the bucket slice when the queue is flushed; autograd gets no gradient from this use
    nbytes = _n.call("spcl_conv_wgrad_workspace_bytes", N, H, W, cin_k, cout_s)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    dw = _grad_buffer(sink, (cout, cin, 3, 3), dev)

    def launch():
        _n.call("spcl_conv3x3_wgrad", _n.ptr(x_store), _n.ptr(dy), dt_code, N, H, W, cin, cin_s, cin_k, cout, cout_s,
                in_mode, _n.ptr(scale), _n.ptr(shift), _n.ptr(ws), _n.ptr(dw), _n.stream())

    if queue is not None and _TAILS:
        # narrow layer straight into a bucket slice: its final sum joins the batched launch (no separate reduce launch)
        if queue.capture_tail(sink, (ws, dy, x_store, scale, shift), launch):
            return None
        return dw
    launch()
    return dw


def _wgrad_up2(x_half, dy, dt_code, N, H, W, cin, cout, cout_s, sink=None):
    """weight gradient of the up-convolution, its input ``nn.Upsample(x2)(x_half)`` read from the half-resolution tensor
    (spcl_conv3x3_wgrad_up2 / spcl_wgrad_item::x_up2); H x W = the convolution's size"""
    dev = dy.device
    queue = sink_queue(sink)
    if queue is not None and _n.call("spcl_conv_wgrad_batched_supported", dt_code, cin, cin, cout, cout_s, 0):
        queue.add(x_half, dy, None, None, sink, N, H, W, cin, cin, cout, cout_s, 0, x_up2=True)
        return None
    nbytes = _n.call("spcl_conv_wgrad_workspace_bytes", N, H, W, cin, cout_s)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    dw = _grad_buffer(sink, (cout, cin, 3, 3), dev)

    def launch():
        _n.call("spcl_conv3x3_wgrad_up2", _n.ptr(x_half), _n.ptr(dy), dt_code, N, H, W, cin, cout, cout_s, _n.ptr(ws),
                _n.ptr(dw), _n.stream())

    if queue is not None and _TAILS:
        if queue.capture_tail(sink, (ws, dy, x_half), launch):
            return None
        return dw
    launch()
    return dw


def up_in_shape_ok(N, cin, H, W, cout, dtype):
    """can ``conv_bn_relu`` read ``nn.Upsample(x2)`` of a dense [N, cin, H/2, W/2] tensor in its loaders (H x W the
    convolution's size)?  (spcl_conv_up2_supported: bf16, sizes the specialised kernels tile)"""
    return bool(_CONV_UP2 and dtype == torch.bfloat16 and cin % 16 == 0
                and _n.call("spcl_conv_up2_supported", _n.dtype_code(dtype), N, H, W, cin, _ru16(cout)))


def _wgrad_cat(xa, xb, dy, dt_code, N, H, W, chalf, cout, cout_s, sink=None, xb_scale=None, xb_shift=None):
    """weight gradient of the convolution of ``cat((xa, xb), channel)``, the input read from the two tensors
    (spcl_conv3x3_wgrad_cat); the final sum rides in the batched launch when ``sink`` belongs to a bucket, as in ``_wgrad``"""
    dev = dy.device
    queue = sink_queue(sink)
    if (queue is not None and xb_scale is None
            and _n.call("spcl_conv_wgrad_batched_supported", dt_code, 2 * chalf, 2 * chalf, cout, cout_s, 0)):
        # a wide layer: one item of the batched launch, its 64-channel input blocks read from the tensor that holds them
        queue.add(xa, dy, None, None, sink, N, H, W, 2 * chalf, 2 * chalf, cout, cout_s, 0, x2=xb)
        return None
    nbytes = _n.call("spcl_conv_wgrad_workspace_bytes", N, H, W, 2 * chalf, cout_s)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    dw = _grad_buffer(sink, (cout, 2 * chalf, 3, 3), dev)

    def launch():
        _n.call("spcl_conv3x3_wgrad_cat", _n.ptr(xa), _n.ptr(xb), _n.ptr(dy), dt_code, N, H, W, chalf, cout, cout_s,
                _n.ptr(xb_scale), _n.ptr(xb_shift), _n.ptr(ws), _n.ptr(dw), _n.stream())

    if queue is not None and _TAILS:
        if queue.capture_tail(sink, (ws, dy, xa, xb, xb_scale, xb_shift), launch):
            return None
        return dw
    launch()
    return dw


_BCAST = os.environ.get("SPCL_POOL_BCAST", "1") != "0"  # A/B switch: 0 materialises the average pool's gradient


def broadcast_rows(g, dtype):
    """``g`` [N, C, H, W] whose every pixel of an (image, channel) holds the same value through stride-0 dimensions (what
    the projector's backward hands over for a global average pool: ``[N, C] .view(N, C, 1, 1).expand(N, C, H, W)``) ->
    the contiguous [N, C] tensor behind it, else None"""
    if (g is None or g.dim() != 4 or g.dtype != dtype or g.shape[1] % 16 or g.shape[2] * g.shape[3] <= 1
            or g.stride(2) != 0 or g.stride(3) != 0 or g.stride(1) != 1 or g.stride(0) != g.shape[1]):
        return None
    return torch.as_strided(g, (g.shape[0], g.shape[1]), (g.shape[1], 1), g.storage_offset())


def _bnrelu_bwd_bcast(y, g_nc, dt_code, dtype, N, H, W, C, cs, st, training, sinks=(None, None)):
    """_bnrelu_bwd for a gradient that is one value per (image, channel) (spcl_bnrelu_backward_bcast)"""
    dev = y.device
    ws = torch.empty(_n.call("spcl_bnrelu_bwd_workspace_bytes", N, H, W, cs) // 4, dtype=torch.float32, device=dev)
    dgamma = _grad_buffer(sinks[0], (C,), dev)
    dbeta = _grad_buffer(sinks[1], (C,), dev)
    dy = torch.empty(N, H, W, cs, dtype=dtype, device=dev)
    _n.call("spcl_bnrelu_backward_bcast", _n.ptr(y), _n.ptr(g_nc), dt_code, N, H, W, C, cs, _n.ptr(st[0]), _n.ptr(st[1]),
            _n.ptr(st[2]), _n.ptr(st[3]), int(training), _n.ptr(ws), _n.ptr(dgamma), _n.ptr(dbeta), _n.ptr(dy), _n.stream())
    return dy, dgamma, dbeta


def _bnrelu_bwd(y, dact, dpool, dt_code, dtype, N, H, W, C, cs, st, training, sinks=(None, None), dact_stride=0, d_up=None):
    """``dact_stride`` > 0: ``dact`` is a channel slice of a wider NHWC tensor with that many elements per pixel.
    ``d_up``: the gradient arrives at twice the resolution ([N, 2H, 2W, cs]: the activation went through nn.Upsample(x2));
    its 2 x 2 sums are formed inside the reduction pass (spcl_bnrelu_backward_up2), ``dact`` is not given"""
    dev = y.device
    ws = torch.empty(_n.call("spcl_bnrelu_bwd_workspace_bytes", N, H, W, cs) // 4, dtype=torch.float32, device=dev)
    dgamma = _grad_buffer(sinks[0], (C,), dev)
    dbeta = _grad_buffer(sinks[1], (C,), dev)
    dy = torch.empty(N, H, W, cs, dtype=dtype, device=dev)
    if d_up is not None:
        assert dact is None and dpool is None
        gsum = torch.empty(N, H, W, cs, dtype=dtype, device=dev)  # (the summed gradient, for the apply pass)
        _n.call("spcl_bnrelu_backward_up2", _n.ptr(y), _n.ptr(d_up), _n.ptr(gsum), dt_code, N, H, W, C, cs, _n.ptr(st[0]),
                _n.ptr(st[1]), _n.ptr(st[2]), _n.ptr(st[3]), int(training), _n.ptr(ws), _n.ptr(dgamma), _n.ptr(dbeta),
                _n.ptr(dy), _n.stream())
    elif dact_stride and dact is not None:
        _n.call("spcl_bnrelu_pool_backward_strided", _n.ptr(y), _n.ptr(dact), int(dact_stride), _n.ptr(dpool), dt_code, N, H,
                W, C, cs, _n.ptr(st[0]), _n.ptr(st[1]), _n.ptr(st[2]), _n.ptr(st[3]), int(training), _n.ptr(ws),
                _n.ptr(dgamma), _n.ptr(dbeta), _n.ptr(dy), _n.stream())
    else:
        _n.call("spcl_bnrelu_pool_backward", _n.ptr(y), _n.ptr(dact), _n.ptr(dpool), dt_code, N, H, W, C, cs,
                _n.ptr(st[0]), _n.ptr(st[1]), _n.ptr(st[2]), _n.ptr(st[3]), int(training), _n.ptr(ws), _n.ptr(dgamma),
                _n.ptr(dbeta), _n.ptr(dy), _n.stream())
    return dy, dgamma, dbeta


def _bnrelu_bwd_image_wgrad(y, dact, image, dt_code, N, H, W, C, cs, st, training, sinks=(None, None, None)):
    """BN+ReLU backward of a one-channel image block's first conv fused with its weight gradient (no dy tensor)."""
    dev = y.device
    nbytes = _n.call("spcl_bnrelu_image_wgrad_workspace_bytes", N, H, W, cs)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    dw = _grad_buffer(sinks[0], (C, 1, 3, 3), dev)
    dgamma = _grad_buffer(sinks[1], (C,), dev)
    dbeta = _grad_buffer(sinks[2], (C,), dev)

    def launch():
        _n.call("spcl_bnrelu_backward_image_wgrad", _n.ptr(y), _n.ptr(dact), _n.ptr(image), dt_code, N, H, W, C, cs,
                _n.ptr(st[0]), _n.ptr(st[1]), _n.ptr(st[2]), _n.ptr(st[3]), int(training), _n.ptr(ws), _n.ptr(dgamma),
                _n.ptr(dbeta), _n.ptr(dw), _n.stream())

    queue = sink_queue(sinks[0])
    if queue is not None and _TAILS:
        if queue.capture_tail(sinks[0], (ws, dact), launch):
            dw = None  # reaches the bucket slice when the queue is flushed
    else:
        launch()
    return dw, dgamma, dbeta


def _image_wgrad_fusable(cfg, cin, cout_s):
    return cfg.image_input and cin == 1 and cout_s <= 256 and cout_s & (cout_s - 1) == 0


def _dgrad_bnstats(dy, wp_t, y2, st2, dt_code, dtype, N, H, W, cin_k, cout_s):
    """input gradient of a block's second conv + the per-tile partial sums of the first conv's BN backward (one kernel)
    -> (g, rows) or None where no specialised kernel exists (the caller then runs the separate reduction pass)."""
    if not _n.call("spcl_conv_dgrad_bnstats_supported", dt_code, N, H, W, cin_k, cout_s):
        return None
    dev = dy.device
    nt = _n.call("spcl_conv_stat_rows", dt_code, N, H, W, cin_k, cout_s)
    g = torch.empty(N, H, W, cout_s, dtype=dtype, device=dev)
    rows = torch.empty(nt * 2 * cout_s, dtype=torch.float32, device=dev)
    _n.call("spcl_conv3x3_dgrad_bnstats", _n.ptr(dy), dt_code, N, H, W, cin_k, cout_s, _n.ptr(wp_t), _n.ptr(g),
            _n.ptr(y2), _n.ptr(st2[2]), _n.ptr(st2[3]), _n.ptr(st2[0]), _n.ptr(rows), _n.stream())
    rows.ntiles = nt
    return g, rows


def _bnrelu_bwd_rows(y, dact, image, rows, dt_code, dtype, N, H, W, C, cs, st, training, sinks):
    """BN+ReLU backward finished from the dgrad's per-tile rows -> (dy or dW, dgamma, dbeta)."""
    dev = y.device
    nbytes = _n.call("spcl_bnrelu_image_wgrad_workspace_bytes" if image is not None else
                     "spcl_bnrelu_bwd_workspace_bytes", N, H, W, cs)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    if image is not None:
        first = _grad_buffer(sinks[0], (C, 1, 3, 3), dev)
        dgamma, dbeta = _grad_buffer(sinks[1], (C,), dev), _grad_buffer(sinks[2], (C,), dev)
        dy, dw = None, first
    else:
        first = torch.empty(N, H, W, cs, dtype=dtype, device=dev)
        dgamma, dbeta = _grad_buffer(sinks[0], (C,), dev), _grad_buffer(sinks[1], (C,), dev)
        dy, dw = first, None
    def launch():
        _n.call("spcl_bnrelu_backward_rows", _n.ptr(y), _n.ptr(dact), _n.ptr(image), _n.ptr(rows), rows.ntiles, dt_code,
                N, H, W, C, cs, _n.ptr(st[0]), _n.ptr(st[1]), _n.ptr(st[2]), _n.ptr(st[3]), int(training), _n.ptr(ws),
                _n.ptr(dgamma), _n.ptr(dbeta), _n.ptr(dy), _n.ptr(dw), _n.stream())

    queue = sink_queue(sinks[0]) if image is not None else None
    if queue is not None and _TAILS:
        if queue.capture_tail(sinks[0], (ws, dact), launch):
            first = None  # the weight gradient reaches the bucket slice when the queue is flushed
    else:
        launch()
    return first, dgamma, dbeta


_IMAGE3 = os.environ.get("SPCL_IMAGE3", "1") != "0"  # A/B switch: 0 keeps the fused BN-backward + weight-gradient pass


def _image3_supported(cfg, cin, dt_code, N, H, W, cout_s):
    """the first conv of a one-channel-image block can get its BN backward and weight gradient WITHOUT a pass over its
    output (csrc/bn.hip image3: dgrad epilogue sums + image autocorrelation)"""
    return bool(_IMAGE3 and cfg.image_input and cin == 1 and cfg.dtype == torch.bfloat16 and W <= 256 and
                _n.call("spcl_conv_dgrad_bnstats_image_supported", dt_code, N, H, W, cout_s, cout_s))


def _image_autocorr(image, N, H, W):
    """partial rows of the 9 x 9 autocorrelation (45 sums) and the 9 shifted sums of the zero-padded f32 image batch"""
    rows = _n.call("spcl_image_autocorr_rows", N, H, W)
    out = torch.empty(rows, 64, dtype=torch.float32, device=image.device)
    _n.call("spcl_image_autocorr", _n.ptr(image), N, H, W, _n.ptr(out), _n.stream())
    return out


def _dgrad_bnstats_image(dy, wp_t, y2, st2, image, dt_code, dtype, N, H, W, cs, want_g=False):
    """_dgrad_bnstats for the image block: the per-tile rows carry nine more sub-rows, sum dz * image[p + tap].  The
    gradient g itself is only written on request: with image3 nothing downstream reads it (103 MB at 64 x 224^2)."""
    dev = dy.device
    nt = _n.call("spcl_conv_stat_rows", dt_code, N, H, W, cs, cs)
    g = torch.empty(N, H, W, cs, dtype=dtype, device=dev) if want_g else None
    rows = torch.empty(nt * 11 * cs, dtype=torch.float32, device=dev)
    _n.call("spcl_conv3x3_dgrad_bnstats_image", _n.ptr(dy), dt_code, N, H, W, cs, cs, _n.ptr(wp_t), _n.ptr(g), _n.ptr(y2),
            _n.ptr(st2[2]), _n.ptr(st2[3]), _n.ptr(st2[0]), _n.ptr(image), _n.ptr(rows), _n.stream())
    rows.ntiles = nt
    return g, rows


_CONV16_FUSED = os.environ.get("SPCL_CONV16_FUSED", "1") != "0"  # A/B switch: 0 = separate wgrad and dgrad launches
_CONV16_WGROWS = os.environ.get("SPCL_CONV16_WGROWS", "1") != "0"  # A/B switch: 0 = per-tile rows + the folding launch


def _conv16_bwd_fused(dy, wp_t, y2, st2, image, dt_code, N, H, W, cin, cout, cs, sink, acorr=None):
    """the image block's second conv, whole backward in one launch (csrc/conv16_bwd.hip): -> (dW or None when it went into
    a bucket slice, rows of the first conv's BatchNorm backward / weight gradient).  With ``acorr`` (the autocorrelation's
    partial rows) the rows come as ONE set per workgroup in the final kernel's layout (``rows.wg``) and the same launch
    folds ``acorr`` to 16 rows (``rows.acorr16``): no per-tile rows, no folding launch; else per tile ([tiles][11][cs])."""
    dev = dy.device
    nsplit = _n.call("spcl_conv16_bwd_fused_splits", N, H, W)
    if acorr is not None:
        rows = torch.empty(11 * cs * nsplit, dtype=torch.float32, device=dev)
        rows.wg, rows.acorr16 = nsplit, torch.empty(16, 64, dtype=torch.float32, device=dev)
    else:
        nt = _n.call("spcl_conv_stat_rows", dt_code, N, H, W, cs, cs)
        rows = torch.empty(nt * 11 * cs, dtype=torch.float32, device=dev)
        rows.ntiles, rows.wg = nt, 0
    ws = torch.empty(nsplit * 9 * 256, dtype=torch.float32, device=dev)
    dw = _grad_buffer(sink, (cout, cin, 3, 3), dev)

    def launch():
        _n.call("spcl_conv16_bwd_fused", _n.ptr(dy), dt_code, N, H, W, _n.ptr(wp_t), _n.ptr(y2), _n.ptr(st2[2]),
                _n.ptr(st2[3]), _n.ptr(st2[0]), _n.ptr(image), _n.ptr(rows) if not rows.wg else None, _n.ptr(ws), _n.ptr(dw),
                cin, cout, _n.ptr(rows) if rows.wg else None, _n.ptr(acorr), 0 if acorr is None else acorr.shape[0],
                _n.ptr(rows.acorr16) if rows.wg else None, _n.stream())

    queue = sink_queue(sink)
    if queue is not None and _TAILS:
        if queue.capture_tail(sink, (ws, dy, y2, image, rows), launch):
            return None, rows
        return dw, rows
    launch()
    return dw, rows


def _bnrelu_bwd_rows_image3(rows, acorr, w, N, H, W, C, cs, st, training, sinks):
    """BN + ReLU backward of the image block's first conv and its weight gradient from the rows and the autocorrelation
    -> (dW, dgamma, dbeta); dW goes straight into its sink (no tail: the final kernel writes it)"""
    dev = rows.device
    ws = torch.empty(_n.call("spcl_bnrelu_image3_workspace_bytes", cs) // 4, dtype=torch.float32, device=dev)
    dw = _grad_buffer(sinks[0], (C, 1, 3, 3), dev)
    dgamma, dbeta = _grad_buffer(sinks[1], (C,), dev), _grad_buffer(sinks[2], (C,), dev)
    wc = w.detach().contiguous().float()
    if getattr(rows, "wg", 0):  # one row set per workgroup of the one-pass kernel, autocorrelation already folded
        _n.call("spcl_bnrelu_backward_wgrows_image3", _n.ptr(rows), rows.wg, _n.ptr(rows.acorr16), 16, _n.ptr(wc), N, H, W, C,
                cs, _n.ptr(st[0]), _n.ptr(st[1]), _n.ptr(st[2]), int(training), _n.ptr(ws), _n.ptr(dgamma), _n.ptr(dbeta),
                _n.ptr(dw), _n.stream())
        return dw, dgamma, dbeta
    _n.call("spcl_bnrelu_backward_rows_image3", _n.ptr(rows), rows.ntiles, _n.ptr(acorr), acorr.shape[0], _n.ptr(wc), N, H,
            W, C, cs, _n.ptr(st[0]), _n.ptr(st[1]), _n.ptr(st[2]), int(training), _n.ptr(ws), _n.ptr(dgamma), _n.ptr(dbeta),
            _n.ptr(dw), _n.stream())
    return dw, dgamma, dbeta


def _bnrelu_pool_bwd_rows(y, dpool, rows, dt_code, dtype, N, H, W, C, cs, st, training, sinks):
    """BN + ReLU + max-pool backward finished from the rows the next block's dgrad left -> (dy, dgamma, dbeta)."""
    dev = y.device
    ws = torch.empty(_n.call("spcl_bnrelu_bwd_workspace_bytes", N, H, W, cs) // 4, dtype=torch.float32, device=dev)
    dgamma = _grad_buffer(sinks[0], (C,), dev)
    dbeta = _grad_buffer(sinks[1], (C,), dev)
    dy = torch.empty(N, H, W, cs, dtype=dtype, device=dev)
    _n.call("spcl_bnrelu_pool_backward_rows", _n.ptr(y), _n.ptr(dpool), _n.ptr(rows), rows.ntiles, dt_code, N, H, W, C, cs,
            _n.ptr(st[0]), _n.ptr(st[1]), _n.ptr(st[2]), _n.ptr(st[3]), int(training), _n.ptr(ws), _n.ptr(dgamma),
            _n.ptr(dbeta), _n.ptr(dy), _n.stream())
    return dy, dgamma, dbeta


ACC_ROWS = object()  # ``PoolLink.rows`` / a dgrad's ``rows`` when the sums went into a fixed-point accumulator block instead


def _bnrelu_bwd_fill(y, dact, dpool, dt_code, dtype, N, H, W, C, cs, st, training, sinks, acc, dact_stride=0, d_up=None):
    """``_bnrelu_bwd`` whose reduction pass ADDS its sums to the zeroed block ``acc`` and whose apply pass derives the
    coefficients from it (spcl_bnrelu_backward_fill_acc): two launches, no finalize launch -> (dy, dgamma, dbeta)"""
    dev = y.device
    dgamma = _grad_buffer(sinks[0], (C,), dev)
    dbeta = _grad_buffer(sinks[1], (C,), dev)
    dy = torch.empty(N, H, W, cs, dtype=dtype, device=dev)
    if d_up is not None:
        assert dact is None and dpool is None
        dact = torch.empty(N, H, W, cs, dtype=dtype, device=dev)  # (the summed gradient, for the apply pass)
    _n.call("spcl_bnrelu_backward_fill_acc", _n.ptr(y), _n.ptr(dact), int(dact_stride), _n.ptr(dpool), _n.ptr(d_up), dt_code,
            N, H, W, C, cs, _n.ptr(st), int(training), _n.ptr(acc), _n.ptr(dgamma), _n.ptr(dbeta), _n.ptr(dy), _n.stream())
    return dy, dgamma, dbeta


def _bnrelu_bwd_rows_acc(y, dact, dpool, rows, dt_code, dtype, N, H, W, C, cs, st, training, sinks, acc):
    """``_bnrelu_bwd_rows`` / ``_bnrelu_pool_bwd_rows`` with the rows added to the zeroed block ``acc`` by one small launch and
    the apply pass deriving its coefficients from it (spcl_bnrelu_backward_rows_acc): no finalize launch -> (dy, dgamma, dbeta)"""
    dev = y.device
    dgamma = _grad_buffer(sinks[0], (C,), dev)
    dbeta = _grad_buffer(sinks[1], (C,), dev)
    dy = torch.empty(N, H, W, cs, dtype=dtype, device=dev)
    _n.call("spcl_bnrelu_backward_rows_acc", _n.ptr(y), _n.ptr(dact), _n.ptr(dpool), _n.ptr(rows), rows.ntiles, dt_code, N, H, W,
            C, cs, _n.ptr(st), int(training), _n.ptr(acc), _n.ptr(dgamma), _n.ptr(dbeta), _n.ptr(dy), _n.stream())
    return dy, dgamma, dbeta


def _take_acc(ctx, name):
    """the backward accumulator block ``ctx.<name>`` -- ONCE: a block is zero only for the first backward after its forward
    (a second backward through the same graph finds None and takes the rows + finalize path)"""
    acc = getattr(ctx, name, None)
    setattr(ctx, name, None)
    return acc


def _dgrad_bnstats_acc(dy, wp_t, y2, st2, dt_code, dtype, N, H, W, cin_k, cout_s, acc):
    """``_dgrad_bnstats`` with the sums ADDED to the block ``acc`` (spcl_conv3x3_dgrad_bnstats_acc) -> g"""
    g = torch.empty(N, H, W, cout_s, dtype=dtype, device=dy.device)
    _n.call("spcl_conv3x3_dgrad_bnstats_acc", _n.ptr(dy), dt_code, N, H, W, cin_k, cout_s, _n.ptr(wp_t), _n.ptr(g), _n.ptr(y2),
            _n.ptr(st2[2]), _n.ptr(st2[3]), _n.ptr(st2[0]), _n.ptr(acc), _n.stream())
    return g


def _bnrelu_bwd_acc(y, dact, dpool, dact_nc, dt_code, dtype, N, H, W, C, cs, st, training, sinks, acc):
    """BN + ReLU (+ max-pool) backward's apply pass with its coefficients derived from the block ``acc``
    (spcl_bnrelu_backward_acc): exactly one of dact / dpool (block already filled by the dgrad that produced the gradient) or
    dact_nc [N, cs] (this call's reduction pass fills it) -> (dy, dgamma, dbeta)"""
    dev = y.device
    dgamma = _grad_buffer(sinks[0], (C,), dev)
    dbeta = _grad_buffer(sinks[1], (C,), dev)
    dy = torch.empty(N, H, W, cs, dtype=dtype, device=dev)
    assert st.is_contiguous() and tuple(st.shape) == (4, cs)
    _n.call("spcl_bnrelu_backward_acc", _n.ptr(y), _n.ptr(dact), _n.ptr(dpool), _n.ptr(dact_nc), dt_code, N, H, W, C, cs,
            _n.ptr(st), int(training), _n.ptr(acc), _n.ptr(dgamma), _n.ptr(dbeta), _n.ptr(dy), _n.stream())
    return dy, dgamma, dbeta


class _ConvBlockFn(torch.autograd.Function):
    """[conv3x3 -> BN -> ReLU] x2 (+ 2x2 max-pool), semi_seg/arch/unet.py:67-82 + :118-121, as HIP kernels.

    Only the raw conv outputs are kept for backward; BN-apply+ReLU of the first conv is fused into the second conv's
    loader, the second's into the (optional) activation / pooled writers."""

    @staticmethod
    def forward(ctx, x, wa, ga, ba, wb, gb, bb, cfg: BlockCfg, x2=None):
        """``x2``: the block's input is ``torch.cat((x, x2), 1)`` (the decoder's skip concatenation, unet.py:194-224), read
        from the two tensors in place (``cat_pair_supported`` must have said yes): no concatenated tensor in the forward,
        ONE gradient tensor in the backward whose channel halves are returned as views."""
        _n.require_gpu(x, wa, wb)
        dtype, dev = cfg.dtype, x.device
        dtc = _n.dtype_code(dtype)
        N, cin, H, W = x.shape
        cout = wa.shape[0]
        cout_s = _ru16(cout)
        x2s = None
        if x2 is not None:
            xs, x2s = to_nhwc_padded(x.detach(), dtype), to_nhwc_padded(x2.detach(), dtype)
            assert not cfg.image_input and x2.shape == x.shape and xs.shape[3] == cin and x2s.shape[3] == cin \
                and wa.shape[1] == 2 * cin, (tuple(x.shape), tuple(x2.shape), tuple(wa.shape))
            chalf, cin = cin, 2 * cin
            cin_s = cin_k = cin
            mode_a = 0
            xl = getattr(cfg, "x2_link", None)
            if xl is not None:  # x2 is the up-convolution's raw output: its BatchNorm + ReLU happens in the loaders
                assert xl.yb.data_ptr() == x2s.data_ptr() and xl.cs == chalf, "x2 is not the linked producer's output"
            ctx.x2_coef = (xl.stb[2], xl.stb[3]) if xl is not None else (None, None)
        elif cfg.image_input:
            if cin > 16:
                raise NotImplementedError("image-input block supports input_dim <= 16")
            xs = x.detach().float().permute(0, 2, 3, 1).contiguous()  # [N,H,W,cin] f32 (a view when cin == 1)
            cin_s, cin_k, mode_a = cin, 16, 2
        else:
            xs = to_nhwc_padded(x.detach(), dtype)
            cin_s = cin_k = xs.shape[3]
            mode_a = 0
        need_bwd = any(ctx.needs_input_grad)
        ctx.acorr = None
        acorr_rows = 0
        if (need_bwd and ctx.needs_input_grad[1] and not ctx.needs_input_grad[0]
                and _image3_supported(cfg, cin, dtc, N, H, W, cout_s)):
            # of the input image only: usually it came with the forward pass's weight-pack launch (UNet._prepack), else now
            ctx.acorr = take_prepacked_acorr(xs)
            acorr_rows = _acorr_in_conv_rows(dtc, N, H, W, cin, cout_s) if (ctx.acorr is None and x2s is None) else 0
            if ctx.acorr is None and acorr_rows == 0:
                ctx.acorr = _image_autocorr(xs, N, H, W)
        pre_a, pre_b = take_prepacked(wa, dtc, H, W), take_prepacked(wb, dtc, H, W)
        if pre_a is not None and pre_b is not None:  # packed with the other layers at the start of the forward pass
            (wpa, wpa_t), (wpb, wpb_t) = pre_a, pre_b
        elif need_bwd:  # the dgrad layouts are packed alongside (same launch) and kept for backward
            (wpa, wpa_t), (wpb, wpb_t) = _pack_block(wa, wb, dtc, dtype, *((H, W) if _PACK_AT else (0, 0)))
        else:
            wpa, wpb, wpa_t, wpb_t = _pack(wa, 0, dtc, dtype), _pack(wb, 0, dtc, dtype), None, None
        ctx.up2 = bool(getattr(cfg, "up2", False)) and cfg.need_act and not cfg.need_pool
        lazy = (bool(getattr(cfg, "lazy_act", False)) and cfg.need_act and not cfg.need_pool and not ctx.up2
                and getattr(cfg, "act_dst", None) is None)
        # BatchNorm sums through fixed-point accumulator blocks where the layer's kernels offer it (bf16, the specialised
        # convolution kernels, few enough tiles: Conv3 .. Conv5 of the 224^2 step): no finalize launch on either BatchNorm
        count = float(N) * H * W
        acc_ok = _BN_ACC and cfg.training and dtype == torch.bfloat16 and cout_s <= 256 and cout == cout_s
        sup = lambda ci, co, kind, out: bool(_n.call("spcl_conv_bn_acc_supported", dtc, N, H, W, ci, co, kind, out))  # noqa: E731
        use_a = (acc_ok and x2s is None and acorr_rows == 0 and mode_a == 0 and sup(cin_k, cout_s, 0, 1)
                 and sup(cout_s, cout_s, 1, 0))
        use_b = (acc_ok and not lazy and not ctx.up2 and getattr(cfg, "act_dst", None) is None
                 and sup(cout_s, cout_s, 1 if use_a else 2, 1))
        keep = []
        ctx.acc_bwd_a = ctx.acc_bwd_b = ctx.acc_rows_a = None
        if use_a:
            acc_a = bn_acc_block(cout_s, dev)
            ya, _ = _conv_acc(xs, dtc, dtype, N, H, W, cin_k, cout_s, wpa, None, None, None, acc_a, False)
            sta = torch.empty(4, cout_s, dtype=torch.float32, device=dev)  # written by the second convolution's first workgroup
            bn_a = _bn_acc_desc(acc_a, cfg, cout, cout_s, ga, ba, 0, count, sta, keep)
        else:
            if x2s is not None:
                ya, sa = _conv_cat(xs, x2s, dtc, dtype, N, H, W, chalf, cout_s, wpa, cfg.training, *ctx.x2_coef)
            elif acorr_rows > 0:  # the image convolution leaves the autocorrelation rows of the image3 backward itself
                ya, sa, ctx.acorr = _conv_image_acorr(xs, dtc, dtype, N, H, W, cin_s, cout_s, wpa, cfg.training, acorr_rows)
            else:
                ya, sa = _conv(xs, dtc, dtype, N, H, W, cin_s, cin_k, cout_s, wpa, mode_a, None, None, cfg.training)
            sta = _bn_stats(sa, cfg, cout, cout_s, ga, ba, 0, dev)
        if use_a or use_b:
            acc_b = bn_acc_block(cout_s, dev) if use_b else None
            yb, sb = _conv_acc(ya, dtc, dtype, N, H, W, cout_s, cout_s, wpb, bn_a if use_a else None,
                               None if use_a else sta[2], None if use_a else sta[3], acc_b, cfg.training)
        else:
            yb, sb = _conv(ya, dtc, dtype, N, H, W, cout_s, cout_s, cout_s, wpb, 1, sta[2], sta[3], cfg.training)
        if use_b:
            stb = torch.empty(4, cout_s, dtype=torch.float32, device=dev)  # written by the activation writer's first workgroup
            bn_b = _bn_acc_desc(acc_b, cfg, cout, cout_s, gb, bb, 1, count, stb, keep)
        else:
            stb = _bn_stats(sb, cfg, cout, cout_s, gb, bb, 1, dev)
        if acc_ok and any(ctx.needs_input_grad):
            # the backward's blocks are zeroed with the forward's (one fill per step): sum dz / sum dz (y - mean) of both
            # BatchNorms, added by whichever kernel produces the incoming gradient
            if not cfg.image_input and _n.call("spcl_conv_dgrad_bnstats_acc_supported", dtc, N, H, W, cout_s, cout_s):
                ctx.acc_bwd_a = bn_acc_block(cout_s, dev)
            elif not cfg.image_input and _ACC_FILL:
                ctx.acc_rows_a = bn_acc_block(cout_s, dev)  # (more tiles than a block takes adds from: the dgrad's ROWS go in)
            ctx.acc_bwd_b = bn_acc_block(cout_s, dev)
        # small maps whose activation is the block's only product (the encoder's last block under a feature tap): the writer
        # also leaves the activation's global average per (image, channel) -- the projector's AdaptiveAvgPool2d((1, 1)) then
        # has nothing to read back (conv_block hangs it on the returned tensor, _ProjectorFn looks for it)
        cfg.gap = None
        want_gap = (_GAP and getattr(cfg, "want_gap", False) and cfg.need_act and not cfg.need_pool and not lazy
                    and not ctx.up2 and cfg.act_dst is None
                    and H * W <= 1024 and _n.call("spcl_bnrelu_gap_supported", dtc, H, W, cout, cout_s))
        if want_gap:
            act, pool = torch.empty(N, H, W, cout_s, dtype=dtype, device=dev), None
            cfg.gap = torch.empty(N, cout, dtype=torch.float32, device=dev)
            _n.call("spcl_bnrelu_gap_forward", _n.ptr(yb), dtc, N, H, W, cout, cout_s, None if use_b else _n.ptr(stb[2]),
                    None if use_b else _n.ptr(stb[3]), ctypes.byref(bn_b) if use_b else None, _n.ptr(act), _n.ptr(cfg.gap),
                    _n.stream())
        elif use_b:
            act = torch.empty(N, H, W, cout_s, dtype=dtype, device=dev) if cfg.need_act else None
            pool = torch.empty(N, H // 2, W // 2, cout_s, dtype=dtype, device=dev) if cfg.need_pool else None
            _n.call("spcl_bnrelu_pool_forward_acc", _n.ptr(yb), dtc, N, H, W, cout_s, ctypes.byref(bn_b), _n.ptr(act),
                    _n.ptr(pool), _n.stream())
        elif lazy:
            # no activation tensor: the consumer (conv1x1_bn) reads yb and applies scale / shift / ReLU in its loader
            act, pool = yb, None
            cfg.link_act = ActLink(yb, stb, N, H, W, cout, cout_s)
        elif ctx.up2:
            act = torch.empty(N, 2 * H, 2 * W, cout_s, dtype=dtype, device=dev)
            pool = None
            _n.call("spcl_bnrelu_up2_forward", _n.ptr(yb), dtc, N, H, W, cout_s, _n.ptr(stb[2]), _n.ptr(stb[3]), _n.ptr(act),
                    _n.stream())
        else:
            act = _act_buffer(cfg, N, H, W, cout_s, dtype, dev) if cfg.need_act else None
            pool = torch.empty(N, H // 2, W // 2, cout_s, dtype=dtype, device=dev) if cfg.need_pool else None
            _bnrelu_fwd(yb, dtc, N, H, W, cout_s, stb[2], stb[3], act, pool)
        ctx.save_for_backward(xs, ya, yb, sta, stb, wa, wb)
        ctx.x2s = x2s  # (not an input of this Function as a tensor object: a detached re-layout, kept directly)
        ctx.params = (wa, ga, ba, wb, gb, bb)
        ctx.packed_t = (wpa_t, wpb_t)
        ctx.cfg = cfg
        if need_bwd and pool is not None and act is None and cfg.training and dtype == torch.bfloat16:
            cfg.link_out = PoolLink(yb, stb, N, H, W, cout_s, ctx.acc_bwd_b)  # the pooled output is this block's only product
        ctx.meta = (N, cin, H, W, cout, cout_s, cin_s, cin_k, mode_a, x.dtype)
        outs = []
        outs.append(nhwc_to_logical(act, cout) if act is not None else None)
        outs.append(nhwc_to_logical(pool, cout) if pool is not None else None)
        return tuple(outs)

    @staticmethod
    def backward(ctx, d_act, d_pool):
        xs, ya, yb, sta, stb, wa, wb = ctx.saved_tensors
        cfg = ctx.cfg
        N, cin, H, W, cout, cout_s, cin_s, cin_k, mode_a, xdt = ctx.meta
        dtype = cfg.dtype
        dtc = _n.dtype_code(dtype)
        g_nc = broadcast_rows(d_act, dtype) if (_BCAST and d_pool is None and d_act is not None and not ctx.up2
                                                  and d_act.shape[1] == cout_s) else None
        d_up = None
        ul = getattr(cfg, "up_link", None)
        if ul is not None and ul.d_up is not None:
            # the consuming up-convolution left its FINE input gradient in the link (UpLink) and sent an unwritten tensor
            # (compared while the link still holds the unwritten tensor: released first, its address is the first thing the
            # allocator hands out again -- to the very re-layout copy a foreign gradient would need)
            du, holder = ul.d_up, ul.holder
            same = (d_act is not None and d_pool is None and d_act.data_ptr() == ul.dx_ptr and d_act.dtype == holder.dtype
                    and holder._version == ul.version and nhwc_channel_slice(d_act, dtype) is None
                    and to_nhwc_padded(d_act, dtype).data_ptr() == ul.dx_ptr)
            dirty = holder._version != ul.version  # (somebody wrote INTO the shared zeros: what they hold now is that gradient)
            ul.d_up, ul.dx_ptr, ul.holder = None, 0, None
            if same:
                d_up, d_act = du, None
            else:
                # somebody else contributed to this activation's gradient: what arrived is THEIR part (the link's tensor
                # holds zeros), the 2 x 2 sums of the up-convolution's fine gradient are added here -- the ordinary kernels
                gs = torch.empty(N, H, W, cout_s, dtype=dtype, device=du.device)
                _n.call("spcl_upsample2x_backward", _n.ptr(du), _n.ptr(gs), dtc, N, H, W, cout_s, _n.stream())
                gl = nhwc_to_logical(gs, cout)
                d_act = gl if d_act is None else d_act.to(gl.dtype) + gl
                if dirty:
                    holder.zero_()  # (restore the shared tensor for its next use)
        elif ctx.up2 and d_act is not None:
            # the forward returned the x2-upsampled activation: its gradient is summed over the 2x2 replicas first -- inside
            # the BatchNorm-backward reduction pass (spcl_bnrelu_backward_up2), or by its own launch
            du = to_nhwc_padded(d_act, dtype)
            if _UP2_BWD_FUSED and d_pool is None:
                d_up, d_act = du, None
            else:
                dsum = torch.empty(N, H, W, cout_s, dtype=dtype, device=du.device)
                _n.call("spcl_upsample2x_backward", _n.ptr(du), _n.ptr(dsum), dtc, N, H, W, cout_s, _n.stream())
                d_act = nhwc_to_logical(dsum, cout)
        da_stride = 0
        da_sl = nhwc_channel_slice(d_act, dtype) if (d_act is not None and g_nc is None) else None
        if da_sl is not None:  # the skip half of a concatenation's gradient, read in place (spcl_bnrelu_pool_backward_strided)
            da_s, da_stride = da_sl
        else:
            da_s = to_nhwc_padded(d_act, dtype) if (d_act is not None and g_nc is None) else None
        dp_s = to_nhwc_padded(d_pool, dtype) if d_pool is not None else None
        if da_s is None and dp_s is None and g_nc is None and d_up is None:
            return (None,) * 9
        x2s = ctx.x2s
        # ---- second conv
        ng = ctx.needs_input_grad
        sk = tuple(take_grad_sink(p, ng[i + 1]) for i, p in enumerate(ctx.params))  # (wa, ga, ba, wb, gb, bb)
        lk = cfg.link_out
        la = getattr(cfg, "link_act", None)
        if la is not None and da_s is not None and dp_s is None and da_stride == 0 and la.fresh(da_s.data_ptr()):
            # the consumer's backward (the 1x1 head) left this BatchNorm's partial sums next to the activation gradient
            dyb, dgb, dbb = _bnrelu_bwd_rows(yb, da_s, None, la.rows, dtc, dtype, N, H, W, cout, cout_s, stb, cfg.training,
                                             sk[4:6])
        elif g_nc is not None:
            # the block's output fed a global average pool only: its gradient is one value per (image, channel)
            acc = _take_acc(ctx, "acc_bwd_b")
            if acc is not None:  # (reduction pass adds into the block, the apply pass derives its coefficients: no finalize)
                dyb, dgb, dbb = _bnrelu_bwd_acc(yb, None, None, g_nc, dtc, dtype, N, H, W, cout, cout_s, stb, cfg.training,
                                                sk[4:6], acc)
            else:
                dyb, dgb, dbb = _bnrelu_bwd_bcast(yb, g_nc, dtc, dtype, N, H, W, cout, cout_s, stb, cfg.training, sk[4:6])
        elif lk is not None and da_s is None and dp_s is not None and lk.fresh(dp_s.data_ptr()):
            # the next block's input-gradient kernel left this BatchNorm's partial sums next to the gradient itself
            if lk.rows is ACC_ROWS:  # ... in this block's accumulator block
                dyb, dgb, dbb = _bnrelu_bwd_acc(yb, None, dp_s, None, dtc, dtype, N, H, W, cout, cout_s, stb, cfg.training,
                                                sk[4:6], _take_acc(ctx, "acc_bwd_b"))
            else:
                acc = _take_acc(ctx, "acc_bwd_b") if (_ACC_FILL and H % 2 == 0 and W % 2 == 0 and cout_s <= 256) else None
                if acc is not None:  # (the rows go into this block's own, untouched, accumulator block: no finalize launch)
                    dyb, dgb, dbb = _bnrelu_bwd_rows_acc(yb, None, dp_s, lk.rows, dtc, dtype, N, H, W, cout, cout_s, stb,
                                                         cfg.training, sk[4:6], acc)
                else:
                    dyb, dgb, dbb = _bnrelu_pool_bwd_rows(yb, dp_s, lk.rows, dtc, dtype, N, H, W, cout, cout_s, stb,
                                                          cfg.training, sk[4:6])
        else:
            acc = _take_acc(ctx, "acc_bwd_b") if _ACC_FILL else None
            if acc is not None and lk is not None and lk.rows is ACC_ROWS:
                # the next block's input-gradient kernel ADDED its sums to this block already -- of a gradient that then got
                # company (a second consumer of the pooled tensor): the block is not zero any more, and its content is not
                # this gradient's
                acc = None
            if acc is not None:
                # a gradient that did not come with its sums (several consumers: the decoder's skip connections): the reduction
                # pass adds them to the block, the apply pass derives its coefficients -- no finalize launch
                dyb, dgb, dbb = _bnrelu_bwd_fill(yb, da_s, dp_s, dtc, dtype, N, H, W, cout, cout_s, stb, cfg.training, sk[4:6],
                                                 acc, dact_stride=da_stride, d_up=d_up)
            else:
                dyb, dgb, dbb = _bnrelu_bwd(yb, da_s, dp_s, dtc, dtype, N, H, W, cout, cout_s, stb, cfg.training, sk[4:6],
                                            dact_stride=da_stride, d_up=d_up)
        if lk is not None:
            lk.rows, lk.dx_ptr, lk.holder = None, 0, None
        if la is not None:
            la.rows, la.dx_ptr, la.holder = None, 0, None
        wpa_t, wpb_t = ctx.packed_t
        if wpb_t is None:
            wpb_t = _pack(wb, 1, dtc, dtype)
        image3 = ctx.acorr is not None
        # image block: this conv's weight gradient AND the sums the first conv's backward needs in one pass over dyb / ya
        one_pass = (image3 and _CONV16_FUSED and ctx.needs_input_grad[4] and cout == cout_s
                    and _n.call("spcl_conv16_bwd_fused_supported", dtc, N, H, W, cout_s, cout_s))
        if one_pass:
            dwb, rows16 = _conv16_bwd_fused(dyb, wpb_t, ya, sta, xs, dtc, N, H, W, cout, cout, cout_s, sk[3],
                                            acorr=ctx.acorr if _CONV16_WGROWS else None)
        else:
            dwb = _wgrad(ya, dyb, dtc, N, H, W, cout, cout_s, cout_s, cout, cout_s, 1, sta[2], sta[3], sk[3]) \
                if ctx.needs_input_grad[4] else None
        # the dgrad's output is d loss / d relu(bn_a(ya)): where a specialised kernel exists its epilogue also leaves the
        # per-tile partial sums of bn_a's backward (no separate reduction pass over ya and the gradient)
        if one_pass:
            fused = (None, rows16)
        elif image3:
            fused = _dgrad_bnstats_image(dyb, wpb_t, ya, sta, xs, dtc, dtype, N, H, W, cout_s)
        else:
            acc = _take_acc(ctx, "acc_bwd_a")
            if acc is not None:  # the sums go into the block, the apply pass below derives its coefficients from it
                fused = (_dgrad_bnstats_acc(dyb, wpb_t, ya, sta, dtc, dtype, N, H, W, cout_s, cout_s, acc), (ACC_ROWS, acc))
            else:
                fused = _dgrad_bnstats(dyb, wpb_t, ya, sta, dtc, dtype, N, H, W, cout_s, cout_s) \
                    if dtype == torch.bfloat16 else None
        if fused is not None:
            daa, rows = fused
        else:
            daa, _ = _conv(dyb, dtc, dtype, N, H, W, cout_s, cout_s, cout_s, wpb_t, 0, None, None, False)
        # ---- first conv
        image_fused = ctx.needs_input_grad[1] and not ctx.needs_input_grad[0] and _image_wgrad_fusable(cfg, cin, cout_s)
        if image3:
            dya = None  # no pass over ya / the gradient at all: rows + image autocorrelation -> dW, dgamma, dbeta
            dwa, dga, dba = _bnrelu_bwd_rows_image3(rows, ctx.acorr, wa, N, H, W, cout, cout_s, sta, cfg.training, sk[0:3])
        elif image_fused:
            dya = None  # dy of this layer feeds only dW: one fused pass, nothing written
            if fused is not None:
                dwa, dga, dba = _bnrelu_bwd_rows(ya, daa, xs, rows, dtc, dtype, N, H, W, cout, cout_s, sta, cfg.training,
                                                 sk[0:3])
            else:
                dwa, dga, dba = _bnrelu_bwd_image_wgrad(ya, daa, xs, dtc, N, H, W, cout, cout_s, sta, cfg.training,
                                                        sk[0:3])
        else:
            if fused is not None and isinstance(rows, tuple) and rows[0] is ACC_ROWS:
                dya, dga, dba = _bnrelu_bwd_acc(ya, daa, None, None, dtc, dtype, N, H, W, cout, cout_s, sta, cfg.training,
                                                sk[1:3], rows[1])
            elif fused is not None:
                acc = _take_acc(ctx, "acc_rows_a")
                if acc is not None:
                    dya, dga, dba = _bnrelu_bwd_rows_acc(ya, daa, None, rows, dtc, dtype, N, H, W, cout, cout_s, sta,
                                                         cfg.training, sk[1:3], acc)
                else:
                    dya, dga, dba = _bnrelu_bwd_rows(ya, daa, None, rows, dtc, dtype, N, H, W, cout, cout_s, sta,
                                                     cfg.training, sk[1:3])
            else:
                dya, dga, dba = _bnrelu_bwd(ya, daa, None, dtc, dtype, N, H, W, cout, cout_s, sta, cfg.training,
                                            sk[1:3])
            if not ctx.needs_input_grad[1]:
                dwa = None
            elif x2s is not None:
                dwa = _wgrad_cat(xs, x2s, dya, dtc, N, H, W, cin // 2, cout, cout_s, sk[0], *ctx.x2_coef)
            else:
                dwa = _wgrad(xs, dya, dtc, N, H, W, cin, cin_s, cin_k, cout, cout_s, mode_a, None, None, sk[0])
        dx = dx2 = None
        if ctx.needs_input_grad[0] or (x2s is not None and ctx.needs_input_grad[8]):
            if cfg.image_input:
                raise NotImplementedError("gradient w.r.t. the input image is not on the hot path")
            if wpa_t is None:
                wpa_t = _pack(wa, 1, dtc, dtype)
            li = cfg.link_in
            dxs = None
            if (li is not None and dtype == torch.bfloat16 and li.cout_s == cin_s and li.N == N
                    and _n.call("spcl_conv_dgrad_poolstats_supported", dtc, N, H, W, cout_s, cin_s, li.H, li.W)):
                dxs = torch.empty(N, H, W, cin_s, dtype=dtype, device=dya.device)
                if (li.acc is not None and _n.call("spcl_conv_dgrad_poolstats_acc_supported", dtc, N, H, W, cout_s, cin_s,
                                                   li.H, li.W)):
                    # ... added to the producing block's accumulator block (zeroed with its forward): no rows, no finalize.
                    # Taken ONCE, like ``_take_acc``: the block is zero only for the first backward after its forward; a second
                    # backward through the same graph (retain_graph=True) finds None here and writes rows
                    acc_in, li.acc = li.acc, None
                    _n.call("spcl_conv3x3_dgrad_poolstats_acc", _n.ptr(dya), dtc, N, H, W, cout_s, cin_s, _n.ptr(wpa_t),
                            _n.ptr(dxs), _n.ptr(li.yb), li.H, li.W, _n.ptr(li.stb[2]), _n.ptr(li.stb[3]), _n.ptr(li.stb[0]),
                            _n.ptr(acc_in), _n.stream())
                    rows = ACC_ROWS
                else:
                    nt = _n.call("spcl_conv_stat_rows", dtc, N, H, W, cout_s, cin_s)
                    rows = torch.empty(nt * 2 * cin_s, dtype=torch.float32, device=dya.device)
                    _n.call("spcl_conv3x3_dgrad_poolstats", _n.ptr(dya), dtc, N, H, W, cout_s, cin_s, _n.ptr(wpa_t),
                            _n.ptr(dxs), _n.ptr(li.yb), li.H, li.W, _n.ptr(li.stb[2]), _n.ptr(li.stb[3]), _n.ptr(li.stb[0]),
                            _n.ptr(rows), _n.stream())
                    rows.ntiles = nt
                li.rows, li.dx_ptr, li.holder, li.version = rows, dxs.data_ptr(), dxs, dxs._version
            split = None
            if (dxs is None and x2s is not None and _CONV_SPLIT
                    and _n.call("spcl_conv_split_supported", dtc, N, H, W, cout_s, cin_s)):
                # the concatenation's gradient as the two dense gradients of its parts (one launch, no interleaved tensor)
                split = (torch.empty(N, H, W, cin_s // 2, dtype=dtype, device=dya.device),
                         torch.empty(N, H, W, cin_s // 2, dtype=dtype, device=dya.device))
                xb = getattr(cfg, "x2_bn", None)
                if (xb is not None and xb.cs == cin_s // 2 and (xb.N, xb.H, xb.W) == (N, H, W) and cfg.training
                        and _n.call("spcl_conv_split_bnstats_supported", dtc, N, H, W, cout_s, cin_s)):
                    # ... and the up-convolution's BatchNorm-backward sums in the same epilogue (its reduction pass disappears)
                    nt = _n.call("spcl_conv_stat_rows", dtc, N, H, W, cout_s, cin_s)
                    rows = torch.empty(nt * 2 * (cin_s // 2), dtype=torch.float32, device=dya.device)
                    rows.ntiles = nt
                    _n.call("spcl_conv3x3_dgrad_split_bnstats", _n.ptr(dya), dtc, N, H, W, cout_s, cin_s, _n.ptr(wpa_t),
                            _n.ptr(split[0]), _n.ptr(split[1]), _n.ptr(xb.yb), _n.ptr(xb.stb[2]), _n.ptr(xb.stb[3]),
                            _n.ptr(xb.stb[0]), _n.ptr(rows), _n.stream())
                    xb.rows, xb.dx_ptr, xb.holder, xb.version = rows, split[1].data_ptr(), split[1], split[1]._version
                else:
                    _n.call("spcl_conv3x3_forward_split", _n.ptr(dya), dtc, N, H, W, cout_s, cin_s, _n.ptr(wpa_t),
                            _n.ptr(split[0]), _n.ptr(split[1]), _n.stream())
            elif dxs is None:
                dxs, _ = _conv(dya, dtc, dtype, N, H, W, cout_s, cout_s, cin_s, wpa_t, 0, None, None, False)
            if x2s is not None:
                if split is not None:
                    dx, dx2 = nhwc_to_logical(split[0], cin // 2), nhwc_to_logical(split[1], cin // 2)
                else:
                    # one gradient tensor for the concatenation; each producer's backward reads its channel half in place
                    # (nhwc_channel_slice -> the _strided BatchNorm entry points)
                    full = dxs.permute(0, 3, 1, 2)
                    dx, dx2 = full[:, :cin // 2], full[:, cin // 2:]
                if dx.dtype != xdt:
                    dx, dx2 = dx.to(xdt), dx2.to(xdt)
                if not ctx.needs_input_grad[0]:
                    dx = None
                if not ctx.needs_input_grad[8]:
                    dx2 = None
            else:
                dx = nhwc_to_logical(dxs, cin)
                if dx.dtype != xdt:
                    dx = dx.to(xdt)
        ng = ctx.needs_input_grad
        return (dx, dwa, dga if ng[2] else None, dba if ng[3] else None, dwb, dgb if ng[5] else None,
                dbb if ng[6] else None, None, dx2)


def conv_block(x, wa, ga, ba, wb, gb, bb, cfg: BlockCfg, x2=None):
    """-> (act or None, pooled or None), logical NCHW views over NHWC storage.  ``x2``: see ``_ConvBlockFn.forward``."""
    act, pool = _ConvBlockFn.apply(x, wa, ga, ba, wb, gb, bb, cfg, x2)
    gap = getattr(cfg, "gap", None)
    if gap is not None and act is not None:
        # the activation's global average, left by its writer: valid while the tensor is not written in place (its version
        # counter says so); a slice / copy of the tensor simply does not carry it and is pooled the ordinary way
        act._spcl_gap = (gap, act._version)
        cfg.gap = None
    return act, pool


def _gap_of(feat, N, C):
    """the [N, C] global average the producer of ``feat`` left on it (``conv_block``), if still valid, else None"""
    gap = getattr(feat, "_spcl_gap", None)
    if (gap is None or gap[1] != feat._version or tuple(gap[0].shape) != (N, C) or gap[0].device != feat.device
            or tuple(feat.shape[:2]) != (N, C)):
        return None
    return gap[0]


def cat_pair_shape_ok(N, C, H, W, cout, dtype):
    """``cat_pair_supported`` for two dense NHWC tensors of this shape that do not exist yet (UNet.forward plans with it)"""
    return bool(_CONV_CAT and dtype == torch.bfloat16
                and _n.call("spcl_conv_cat_supported", _n.dtype_code(dtype), N, H, W, C, _ru16(cout)))


def cat_pair_supported(a, b, cout, dtype):
    """can ``conv_block(a, ..., x2=b)`` stand for ``conv_block(torch.cat((a, b), 1), ...)``?  (two dense bf16 NHWC tensors of
    16 or 32 channels each at a size the specialised convolution kernels tile: spcl_conv_cat_supported)"""
    if not (_CONV_CAT and a.is_cuda and b.is_cuda and a.shape == b.shape and a.dim() == 4 and dtype == torch.bfloat16
            and a.dtype == dtype and b.dtype == dtype):
        return False
    N, C, H, W = a.shape
    for t in (a, b):
        st, cs = as_nhwc(t)
        if cs != C or (st.data_ptr() % 16):
            return False
    return bool(_n.call("spcl_conv_cat_supported", _n.dtype_code(dtype), N, H, W, C, _ru16(cout)))


# --------------------------------------------------------------------------------------------- decoder / head (N1)
class _ConvBNReLUFn(torch.autograd.Function):
    """conv3x3 -> BN -> ReLU, ONCE (the conv of ``_UpConv``, semi_seg/arch/unet.py:85-97, after its nearest upsample),
    from the same kernels as the two-conv block: conv with statistics epilogue, finalize, BN-apply+ReLU writer;
    backward = BN-ReLU backward (2 passes), wgrad, dgrad."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, cfg: BlockCfg):
        _n.require_gpu(x, w)
        dtype, dev = cfg.dtype, x.device
        dtc = _n.dtype_code(dtype)
        N, cin, H, W = x.shape
        up_in = bool(getattr(cfg, "up_in", False))
        if up_in:
            H, W = 2 * H, 2 * W  # the convolution's size: x is the half-resolution tensor, upsampled by the loaders
        cout = w.shape[0]
        cout_s = _ru16(cout)
        xs = to_nhwc_padded(x.detach(), dtype)
        cin_s = xs.shape[3]
        need_bwd = any(ctx.needs_input_grad)
        pre = take_prepacked(w, dtc, H, W)
        if pre is not None:
            wp, wp_t = pre
        elif need_bwd:
            wp, wp_t = _pack_both(w, dtc, dtype)
        else:
            wp, wp_t = _pack(w, 0, dtc, dtype), None
        if up_in:
            assert cin_s == cin, "the in-place upsample needs an unpadded channel count"
            y = torch.empty(N, H, W, cout_s, dtype=dtype, device=dev)
            s = None
            if cfg.training:
                nt = _n.call("spcl_conv_stat_rows", dtc, N, H, W, cin_s, cout_s)
                s = torch.empty(_n.call("spcl_bn_stats_elems", nt, cout_s), dtype=torch.float32, device=dev)
                s.ntiles = nt
            _n.call("spcl_conv3x3_forward_up2", _n.ptr(xs), dtc, N, H, W, cin_s, cout_s, _n.ptr(wp), _n.ptr(y), _n.ptr(s),
                    _n.stream())
        else:
            y, s = _conv(xs, dtc, dtype, N, H, W, cin_s, cin_s, cout_s, wp, 0, None, None, cfg.training)
        st = _bn_stats(s, cfg, cout, cout_s, gamma, beta, 0, dev)
        if bool(getattr(cfg, "lazy_act", False)) and getattr(cfg, "act_dst", None) is None and cout == cout_s:
            # the only consumer (the next block's two-tensor convolution) applies this BatchNorm + ReLU in its loaders
            act = y
            cfg.link_act = ActLink(y, st, N, H, W, cout, cout_s)
        else:
            act = _act_buffer(cfg, N, H, W, cout_s, dtype, dev)
            _bnrelu_fwd(y, dtc, N, H, W, cout_s, st[2], st[3], act, None)
        ctx.save_for_backward(xs, y, st, w)
        ctx.params = (w, gamma, beta)
        ctx.packed_t = wp_t
        ctx.cfg = cfg
        # a block of the step's (zeroed) accumulator arena for the backward's sums: reduction pass adds, apply pass derives
        ctx.acc_bwd = (bn_acc_block(cout_s, dev) if (_BN_ACC and _ACC_FILL and need_bwd and cfg.training and cout_s <= 256
                                                     and dtype == torch.bfloat16) else None)
        cfg.bn_link = ActLink(y, st, N, H, W, cout, cout_s) if (need_bwd and cout == cout_s) else None
        ctx.meta = (N, cin, H, W, cout, cout_s, cin_s, x.dtype)
        ctx.up_in = up_in
        return nhwc_to_logical(act, cout)

    @staticmethod
    def backward(ctx, d_act):
        xs, y, st, w = ctx.saved_tensors
        cfg = ctx.cfg
        N, cin, H, W, cout, cout_s, cin_s, xdt = ctx.meta
        up_in = ctx.up_in
        dtype = cfg.dtype
        dtc = _n.dtype_code(dtype)
        da_sl = nhwc_channel_slice(d_act, dtype)
        da_s, da_stride = da_sl if da_sl is not None else (to_nhwc_padded(d_act, dtype), 0)
        ng = ctx.needs_input_grad
        sk = tuple(take_grad_sink(p, ng[i + 1]) for i, p in enumerate(ctx.params))
        bl = getattr(cfg, "bn_link", None)
        if bl is not None and da_stride == 0 and bl.fresh(da_s.data_ptr()):
            # the consumer's input-gradient kernel left this BatchNorm's backward sums next to the gradient itself
            dy, dg, db = _bnrelu_bwd_rows(y, da_s, None, bl.rows, dtc, dtype, N, H, W, cout, cout_s, st, cfg.training, sk[1:3])
        else:
            acc = _take_acc(ctx, "acc_bwd")
            if acc is not None:
                dy, dg, db = _bnrelu_bwd_fill(y, da_s, None, dtc, dtype, N, H, W, cout, cout_s, st, cfg.training, sk[1:3], acc,
                                              dact_stride=da_stride)
            else:
                dy, dg, db = _bnrelu_bwd(y, da_s, None, dtc, dtype, N, H, W, cout, cout_s, st, cfg.training, sk[1:3],
                                         dact_stride=da_stride)
        if bl is not None:
            bl.rows, bl.dx_ptr, bl.holder = None, 0, None
        if not ctx.needs_input_grad[1]:
            dw = None
        elif up_in:
            dw = _wgrad_up2(xs, dy, dtc, N, H, W, cin, cout, cout_s, sk[0])
        else:
            dw = _wgrad(xs, dy, dtc, N, H, W, cin, cin_s, cin_s, cout, cout_s, 0, None, None, sk[0])
        dx = None
        if ctx.needs_input_grad[0]:
            wp_t = ctx.packed_t if ctx.packed_t is not None else _pack(w, 1, dtc, dtype)
            dxs, _ = _conv(dy, dtc, dtype, N, H, W, cout_s, cout_s, cin_s, wp_t, 0, None, None, False)
            if up_in:  # the gradient w.r.t. the half-resolution input: the 2 x 2 sums of the fine gradient
                ul = getattr(cfg, "up_link", None)
                if ul is not None and _UP2_BWD_FUSED and cfg.training:
                    # the producing block forms the sums (UpLink); what travels through autograd in their place is a shared
                    # tensor of ZEROS (one per shape, never written: the cache's reference keeps autograd from accumulating
                    # into it in place) -- a second consumer of the activation then adds its gradient to zeros, and the block,
                    # which sees a tensor that is not this one, adds the sums to that (it used to be unwritten
                    # memory, and the block could only refuse)
                    dsum = _up_zeros(N, H // 2, W // 2, cin_s, dtype, dxs.device)
                    ul.d_up, ul.dx_ptr, ul.holder, ul.version = dxs, dsum.data_ptr(), dsum, dsum._version
                else:
                    dsum = torch.empty(N, H // 2, W // 2, cin_s, dtype=dtype, device=dxs.device)
                    _n.call("spcl_upsample2x_backward", _n.ptr(dxs), _n.ptr(dsum), dtc, N, H // 2, W // 2, cin_s, _n.stream())
                dxs = dsum
            dx = nhwc_to_logical(dxs, cin)
            if dx.dtype != xdt:
                dx = dx.to(xdt)
        ng = ctx.needs_input_grad
        return dx, dw, dg if ng[2] else None, db if ng[3] else None, None


_UP_ZEROS = {}


def _up_zeros(N, H, W, cs, dtype, device):
    """the shared all-zero [N, H, W, cs] tensor an UpLink sends through autograd in place of the gradient sums it hands over"""
    key = (N, H, W, cs, dtype, device.type, device.index)
    t = _UP_ZEROS.get(key)
    if t is None:
        t = _UP_ZEROS[key] = torch.zeros(N, H, W, cs, dtype=dtype, device=device)
    return t


def conv_bn_relu(x, w, gamma, beta, cfg: BlockCfg):
    return _ConvBNReLUFn.apply(x, w, gamma, beta, cfg)


def _class_map_storage(t: torch.Tensor):
    """logical [N,K,H,W] f32 -> contiguous [N,H,W,K] storage (zero-copy for the channels-last views these ops return)."""
    assert t.dim() == 4
    if t.dtype != torch.float32:
        t = t.float()
    s = t.permute(0, 2, 3, 1)
    return s if s.is_contiguous() else s.contiguous()


class _Conv1x1Fn(torch.autograd.Function):
    """nn.Conv2d(C, K, 1) with bias (``_Deconv_1x1``, unet.py:147,229) -> f32 class map, logical [N,K,H,W]."""

    @staticmethod
    def forward(ctx, x, w, b, dtype):
        _n.require_gpu(x, w, b)
        N, C, H, W = x.shape
        K = w.shape[0]
        if K > 16 or C > 256:
            raise NotImplementedError("conv1x1 head: at most 16 classes / 256 input channels")
        xs = to_nhwc_padded(x.detach(), dtype)
        cs = xs.shape[3]
        wc, bc = w.detach().reshape(K, C).contiguous().float(), b.detach().contiguous().float()
        out = torch.empty(N, H, W, K, dtype=torch.float32, device=x.device)
        _n.call("spcl_conv1x1_forward", _n.ptr(xs), _n.dtype_code(dtype), N * H * W, C, cs, K, _n.ptr(wc), _n.ptr(bc),
                _n.ptr(out), _n.stream())
        ctx.save_for_backward(xs, wc)
        ctx.params = (w, b)
        ctx.meta = (N, C, H, W, K, cs, dtype, x.dtype, tuple(w.shape))
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dout):
        xs, wc = ctx.saved_tensors
        N, C, H, W, K, cs, dtype, xdt, wshape = ctx.meta
        do = _class_map_storage(dout)
        dev = do.device
        dxs = torch.empty(N, H, W, cs, dtype=dtype, device=dev)
        sinks = (take_grad_sink(ctx.params[0], ctx.needs_input_grad[1]),
                 take_grad_sink(ctx.params[1], ctx.needs_input_grad[2]))
        dw = _grad_buffer(sinks[0], (K, C), dev)
        db = _grad_buffer(sinks[1], (K,), dev)
        ws = torch.empty(_n.call("spcl_conv1x1_bwd_workspace_bytes", C, K) // 4, dtype=torch.float32, device=dev)
        _n.call("spcl_conv1x1_backward", _n.ptr(xs), _n.ptr(do), _n.dtype_code(dtype), N * H * W, C, cs, K, _n.ptr(wc),
                _n.ptr(dxs), _n.ptr(dw), _n.ptr(db), _n.ptr(ws), _n.stream())
        dx = nhwc_to_logical(dxs, C)
        if dx.dtype != xdt:
            dx = dx.to(xdt)
        return dx, dw.reshape(wshape), db, None


def conv1x1(x, w, b, dtype):
    return _Conv1x1Fn.apply(x, w, b, dtype)


class _Conv1x1BnFn(torch.autograd.Function):
    """``_Deconv_1x1(relu(bn(y)))`` for the block that returned its RAW second-conv output ``y`` (``BlockCfg.lazy_act``,
    ``link``: its ``ActLink``): BN + ReLU in the loader of the 1x1 convolution (spcl_conv1x1_forward_bn); the backward
    returns the gradient w.r.t. the ACTIVATION -- what the block's backward expects -- and leaves the BatchNorm-backward
    partial sums in the link (spcl_conv1x1_backward_bn)."""

    @staticmethod
    def forward(ctx, y, w, b, link: ActLink):
        _n.require_gpu(y, w, b)
        N, C, H, W = y.shape
        K = w.shape[0]
        if K > 16 or C > 256:
            raise NotImplementedError("conv1x1 head: at most 16 classes / 256 input channels")
        ys = link.yb
        assert (N, H, W, C) == (link.N, link.H, link.W, link.C) and ys.data_ptr() == y.data_ptr(), "not the linked block's output"
        cs, dtype = link.cs, ys.dtype
        wc, bc = w.detach().reshape(K, C).contiguous().float(), b.detach().contiguous().float()
        out = torch.empty(N, H, W, K, dtype=torch.float32, device=y.device)
        _n.call("spcl_conv1x1_forward_bn", _n.ptr(ys), _n.dtype_code(dtype), N * H * W, C, cs, K, _n.ptr(link.stb[2]),
                _n.ptr(link.stb[3]), _n.ptr(wc), _n.ptr(bc), _n.ptr(out), _n.stream())
        ctx.save_for_backward(wc)
        ctx.link = link
        ctx.params = (w, b)
        ctx.meta = (N, C, H, W, K, cs, dtype, y.dtype, tuple(w.shape))
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dout):
        (wc,) = ctx.saved_tensors
        link = ctx.link
        N, C, H, W, K, cs, dtype, ydt, wshape = ctx.meta
        do = _class_map_storage(dout)
        dev = do.device
        dact = torch.empty(N, H, W, cs, dtype=dtype, device=dev)
        sinks = (take_grad_sink(ctx.params[0], ctx.needs_input_grad[1]),
                 take_grad_sink(ctx.params[1], ctx.needs_input_grad[2]))
        dw = _grad_buffer(sinks[0], (K, C), dev)
        db = _grad_buffer(sinks[1], (K,), dev)
        ws = torch.empty(_n.call("spcl_conv1x1_bwd_workspace_bytes", C, K) // 4, dtype=torch.float32, device=dev)
        nrows = _n.call("spcl_conv1x1_bwd_rows", N * H * W)
        rows = torch.empty(nrows * 2 * cs, dtype=torch.float32, device=dev)
        rows.ntiles = nrows
        st = link.stb
        _n.call("spcl_conv1x1_backward_bn", _n.ptr(link.yb), _n.ptr(do), _n.dtype_code(dtype), N * H * W, C, cs, K,
                _n.ptr(st[2]), _n.ptr(st[3]), _n.ptr(st[0]), _n.ptr(wc), _n.ptr(dact), _n.ptr(dw), _n.ptr(db), _n.ptr(ws),
                _n.ptr(rows), _n.stream())
        link.rows, link.dx_ptr, link.holder, link.version = rows, dact.data_ptr(), dact, dact._version
        dx = nhwc_to_logical(dact, C)
        if dx.dtype != ydt:
            dx = dx.to(ydt)
        return dx, dw.reshape(wshape), db, None


def conv1x1_bn(y, w, b, link: ActLink):
    return _Conv1x1BnFn.apply(y, w, b, link)


class _SoftmaxFn(torch.autograd.Function):
    """``logits.softmax(1)`` on a logical [N,K,H,W] class map (new_epocher.py:86,271)."""

    @staticmethod
    def forward(ctx, logits):
        _n.require_gpu(logits)
        ls = _class_map_storage(logits.detach())
        N, H, W, K = ls.shape
        prob = torch.empty_like(ls)
        _n.call("spcl_softmax_forward", _n.ptr(ls), N * H * W, K, _n.ptr(prob), _n.stream())
        ctx.save_for_backward(prob)
        return prob.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dprob):
        (prob,) = ctx.saved_tensors
        N, H, W, K = prob.shape
        dp = _class_map_storage(dprob)
        dl = torch.empty_like(prob)
        _n.call("spcl_softmax_backward", _n.ptr(prob), _n.ptr(dp), N * H * W, K, _n.ptr(dl), _n.stream())
        return dl.permute(0, 3, 1, 2)


def softmax_classes(logits):
    return _SoftmaxFn.apply(logits)


class _KLDivFn(torch.autograd.Function):
    """deepclustering2.loss.KL_div(reduction='mean')(prob, target): mean over positions of
    sum_c -target log((prob+eps)/(target+eps)); gradient w.r.t. ``prob`` only (as the reference uses it)."""

    @staticmethod
    def forward(ctx, prob, target, eps):
        _n.require_gpu(prob, target)
        ps, ts = _class_map_storage(prob.detach()), _class_map_storage(target.detach())
        N, H, W, K = ps.shape
        dev = ps.device
        ws = torch.empty(_n.call("spcl_kl_workspace_bytes") // 4, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        _n.call("spcl_kl_div_forward", _n.ptr(ps), _n.ptr(ts), N * H * W, K, c_float(eps), _n.ptr(ws), _n.ptr(loss),
                _n.stream())
        ctx.save_for_backward(ps, ts)
        ctx.eps = eps
        return loss

    @staticmethod
    def backward(ctx, g):
        ps, ts = ctx.saved_tensors
        N, H, W, K = ps.shape
        gs = g.detach().reshape(1).float().contiguous()
        dp = torch.empty_like(ps)
        _n.call("spcl_kl_div_backward", _n.ptr(ps), _n.ptr(ts), N * H * W, K, c_float(ctx.eps), _n.ptr(gs), _n.ptr(dp),
                _n.stream())
        return dp.permute(0, 3, 1, 2), None, None


def kl_div(prob, target, eps=1e-16):
    return _KLDivFn.apply(prob, target, eps)


class _SupLossFn(torch.autograd.Function):
    """``KL_div(logits.softmax(1), class2one_hot(target, C))`` + the Dice counts of ``logits.max(1)[1]`` against ``target``
    (semi_seg/epochers/new_epocher.py:268-282) in ONE launch over the class map; the gradient w.r.t. the logits for a unit
    upstream gradient is written by the same launch and scaled in backward.  ``counts`` receives (inter, union) [B, C]."""

    @staticmethod
    def forward(ctx, logits, labels, eps, counts):
        _n.require_gpu(logits, labels)
        ls = _class_map_storage(logits.detach())
        N, H, W, K = ls.shape
        lab = labels.detach().long().contiguous()
        if tuple(lab.shape) != (N, H, W):
            raise AssertionError(f"labels {tuple(lab.shape)} do not match the class map {(N, H, W)}")
        dev = ls.device
        ws = torch.empty(_n.call("spcl_kl_workspace_bytes") // 4, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        dl = torch.empty_like(ls)
        both = torch.empty(2, N, K, dtype=torch.int64, device=dev)  # (written by the launch; a caller that keeps them clones ``inter._base`` once)
        inter, union = both[0], both[1]
        _n.call("spcl_sup_loss_forward", _n.ptr(ls), _n.ptr(lab), N, H * W, K, c_float(eps), _n.ptr(ws), _n.ptr(loss),
                _n.ptr(dl), _n.ptr(inter), _n.ptr(union), _n.stream())
        counts.append((inter, union))
        ctx.save_for_backward(dl)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        if is_unit_gradient(g):  # the epocher's own ``backward(gradient=ones)``: x * 1.0 is x, the pass over the class map is skipped
            return dl.permute(0, 3, 1, 2), None, None, None
        return (dl * g.detach().float()).permute(0, 3, 1, 2), None, None, None


_UNIT_GRADIENTS = {}  # data_ptr -> weakref of a 0-dim tensor its owner promises to keep at exactly 1.0


def register_unit_gradient(t: torch.Tensor):
    """``t`` (0-dim, value 1.0, never written again by its owner) is what ``loss.backward(gradient=t)`` will be called with:
    a backward that recognises it by its storage skips the multiplication by it.  Held weakly."""
    import weakref
    assert t.numel() == 1
    _UNIT_GRADIENTS[t.data_ptr()] = (weakref.ref(t), t._version)
    return t


def is_unit_gradient(g: torch.Tensor) -> bool:
    ent = _UNIT_GRADIENTS.get(g.data_ptr()) if g.numel() == 1 else None
    if ent is None:
        return False
    ref, version = ent
    t = ref()
    if t is None or t.data_ptr() != g.data_ptr() or t.dtype != g.dtype:
        _UNIT_GRADIENTS.pop(g.data_ptr(), None)  # (the registered tensor is gone: its address may belong to anything now)
        return False
    if t._version != version:
        # written in place since it was registered (a caller scaling the loss through it): no longer known to hold 1.0 -- the
        # promise is broken for good, the backward multiplies by whatever it holds now
        _UNIT_GRADIENTS.pop(g.data_ptr(), None)
        return False
    return True


def sup_loss_kl_onehot(logits, labels, eps=1e-16):
    """-> (loss, (inter, union)): the fine-tune criterion and the training batch's Dice counts, fused (``_SupLossFn``)"""
    counts = []
    loss = _SupLossFn.apply(logits, labels, eps, counts)
    return loss, counts[0]


def one_hot_classes(labels: torch.Tensor, K: int) -> torch.Tensor:
    """class2one_hot: [N,H,W] integer labels -> logical [N,K,H,W] f32 one-hot (NHWC storage)."""
    _n.require_gpu(labels)
    lab = labels.detach().long().contiguous()
    N, H, W = lab.shape
    out = torch.empty(N, H, W, K, dtype=torch.float32, device=lab.device)
    _n.call("spcl_one_hot", _n.ptr(lab), N * H * W, K, _n.ptr(out), _n.stream())
    return out.permute(0, 3, 1, 2)


def argmax_classes(logits: torch.Tensor) -> torch.Tensor:
    """``logits.max(1)[1]`` -> [N,H,W] int64 (first maximum)."""
    _n.require_gpu(logits)
    ls = _class_map_storage(logits.detach())
    N, H, W, K = ls.shape
    out = torch.empty(N, H, W, dtype=torch.int64, device=ls.device)
    _n.call("spcl_argmax_classes", _n.ptr(ls), N * H * W, K, _n.ptr(out), _n.stream())
    return out


def dice_counts(pred: torch.Tensor, target: torch.Tensor, C: int):
    """per sample and class: intersection and union counts of two class-coded maps -> two [B,C] int64 tensors."""
    _n.require_gpu(pred, target)
    if pred.shape != target.shape:
        raise AssertionError(f"incompatible shape of `pred` and `target`, given {pred.shape} and {target.shape}.")
    p, t = pred.detach().long().contiguous(), target.detach().long().contiguous()
    B = p.shape[0]
    inter = torch.zeros(B, C, dtype=torch.int64, device=p.device)
    union = torch.zeros(B, C, dtype=torch.int64, device=p.device)
    _n.call("spcl_dice_counts", _n.ptr(p), _n.ptr(t), B, p[0].numel(), C, _n.ptr(inter), _n.ptr(union), _n.stream())
    return inter, union


class _Upsample2xFn(torch.autograd.Function):
    """nn.Upsample(scale_factor=2) (nearest) of ``_UpConv`` on NHWC storage; backward = 2x2 sum."""

    @staticmethod
    def forward(ctx, x, dtype):
        _n.require_gpu(x)
        N, C, H, W = x.shape
        xs = to_nhwc_padded(x.detach(), dtype)
        cs = xs.shape[3]
        y = torch.empty(N, 2 * H, 2 * W, cs, dtype=dtype, device=x.device)
        _n.call("spcl_upsample2x_forward", _n.ptr(xs), _n.ptr(y), _n.dtype_code(dtype), N, H, W, cs, _n.stream())
        ctx.meta = (N, C, H, W, cs, dtype, x.dtype)
        return nhwc_to_logical(y, C)

    @staticmethod
    def backward(ctx, dy):
        N, C, H, W, cs, dtype, xdt = ctx.meta
        dys = to_nhwc_padded(dy, dtype)
        dx = torch.empty(N, H, W, cs, dtype=dtype, device=dy.device)
        _n.call("spcl_upsample2x_backward", _n.ptr(dys), _n.ptr(dx), _n.dtype_code(dtype), N, H, W, cs, _n.stream())
        out = nhwc_to_logical(dx, C)
        return (out if out.dtype == xdt else out.to(xdt)), None


def upsample2x(x, dtype):
    return _Upsample2xFn.apply(x, dtype)


class _Concat2Fn(torch.autograd.Function):
    """torch.cat((a, b), dim=1) of two logical-NCHW tensors with NHWC storage (channel counts multiples of 16);
    backward = one split launch into two contiguous gradients."""

    @staticmethod
    def forward(ctx, a, b, dtype):
        _n.require_gpu(a, b)
        N, CA, H, W = a.shape
        CB = b.shape[1]
        sa, sb = to_nhwc_padded(a.detach(), dtype), to_nhwc_padded(b.detach(), dtype)
        out = torch.empty(N, H, W, CA + CB, dtype=dtype, device=a.device)
        _n.call("spcl_concat2_channels", _n.ptr(sa), _n.ptr(sb), _n.ptr(out), out.element_size(), N * H * W, CA, CB,
                _n.stream())
        ctx.meta = (N, CA, CB, H, W, dtype, a.dtype, b.dtype)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        N, CA, CB, H, W, dtype, adt, bdt = ctx.meta
        gs = to_nhwc_padded(g, dtype)
        ga = torch.empty(N, H, W, CA, dtype=dtype, device=g.device)
        gb = torch.empty(N, H, W, CB, dtype=dtype, device=g.device)
        _n.call("spcl_split2_channels", _n.ptr(gs), _n.ptr(ga), _n.ptr(gb), gs.element_size(), N * H * W, CA, CB,
                _n.stream())
        ga, gb = ga.permute(0, 3, 1, 2), gb.permute(0, 3, 1, 2)
        return (ga if ga.dtype == adt else ga.to(adt)), (gb if gb.dtype == bdt else gb.to(bdt)), None


class _VirtualCatFn(torch.autograd.Function):
    """``torch.cat((a, b), dim=1)`` (semi_seg/arch/unet.py:194-224) when a and b ALREADY are the two channel halves of one
    dense NHWC buffer -- their producers wrote them there (``BlockCfg.act_dst``): no launch forward, no launch backward (the
    gradient's halves are handed on as channel-slice views, which the BatchNorm-backward kernels read in place)."""

    @staticmethod
    def forward(ctx, a, b, holder):
        buf = holder[0]
        ctx.ca = a.shape[1]
        return buf.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        return g[:, :ctx.ca], g[:, ctx.ca:], None


def cat_buffer(N, H, W, ca, cb, dtype, device):
    """a dense [N,H,W,ca+cb] buffer and its two channel-slice views (destinations for the producers of a concatenation)"""
    buf = torch.empty(N, H, W, ca + cb, dtype=dtype, device=device)
    return buf, buf[..., :ca], buf[..., ca:]


def virtual_cat(a, b, buf):
    """cat((a, b), 1) without a copy when a and b are the two halves of ``buf`` (see ``cat_buffer``); else the copying kernel"""
    ca, cb = a.shape[1], b.shape[1]
    es = buf.element_size()
    ok = (a.dtype == b.dtype == buf.dtype and a.stride(1) == 1 and b.stride(1) == 1
          and a.data_ptr() == buf.data_ptr() and b.data_ptr() == buf.data_ptr() + ca * es
          and a.stride(3) == ca + cb == b.stride(3) and tuple(a.shape[2:]) == tuple(buf.shape[1:3]) == tuple(b.shape[2:])
          and a.shape[0] == buf.shape[0] == b.shape[0] and buf.shape[3] == ca + cb)
    if not ok:
        return concat_channels(a, b, buf.dtype)
    return _VirtualCatFn.apply(a, b, [buf])


def concat_channels(a, b, dtype):
    """cat((a, b), 1); channel counts that are not multiples of 16 (tiny test networks) go through torch.cat."""
    if a.shape[1] % 16 or b.shape[1] % 16:
        return torch.cat((a, b), dim=1)
    return _Concat2Fn.apply(a, b, dtype)
