"""torch.autograd.Function wrappers around the C ABI (native.py).  Device tensors in, device tensors out;
no host synchronisation anywhere in this file."""
from __future__ import annotations

from ctypes import c_float, c_int

import torch

from . import native as _n

SP_NONE, SP_HARD, SP_SOFT = 0, 1, 2


# --------------------------------------------------------------------------------------------- contrastive loss
class SupConState:
    """Device-side results of one loss evaluation (kept for backward and for the lazily materialised taps)."""
    __slots__ = ("n", "d", "t", "sp_mode", "gamma", "labels", "mask", "ws", "out")


class _SupConFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z1, z2, labels, mask, t, sp_mode, gamma, correct_grad, state: SupConState):
        _n.require_gpu(z1, z2, labels, mask)
        z1c = z1.detach().contiguous().float()
        z2c = z2.detach().contiguous().float()
        n, d = z1c.shape
        nbytes = _n.call("spcl_supcon_workspace_bytes", n, d)
        if nbytes == 0:
            raise RuntimeError(f"supcon: unsupported shape n={n} d={d} (d must be <= 256)")
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=z1.device)
        out = torch.zeros(8, dtype=torch.float32, device=z1.device)
        _n.call("spcl_supcon_forward", _n.ptr(z1c), _n.ptr(z2c), _n.ptr(labels), _n.ptr(mask), n, d, c_float(t),
                sp_mode, c_float(gamma), int(bool(correct_grad)), _n.ptr(ws), _n.ptr(out), _n.stream())
        state.n, state.d, state.t, state.sp_mode, state.gamma = n, d, t, sp_mode, gamma
        state.labels, state.mask, state.ws, state.out = labels, mask, ws, out
        ctx.state = state
        ctx.in_dtypes = (z1.dtype, z2.dtype)
        return out[0].clone()

    @staticmethod
    def backward(ctx, grad_out):
        s = ctx.state
        dev = s.ws.device
        dz1 = torch.empty(s.n, s.d, dtype=torch.float32, device=dev)
        dz2 = torch.empty(s.n, s.d, dtype=torch.float32, device=dev)
        wsb = torch.empty(_n.call("spcl_supcon_bwd_workspace_bytes", s.n, s.d) // 4, dtype=torch.float32, device=dev)
        go = grad_out.detach().reshape(1).float().contiguous()
        _n.call("spcl_supcon_backward", _n.ptr(s.labels), _n.ptr(s.mask), s.n, s.d, c_float(s.t), s.sp_mode,
                c_float(s.gamma), _n.ptr(s.ws), _n.ptr(wsb), _n.ptr(s.out), _n.ptr(go), _n.ptr(dz1), _n.ptr(dz2),
                _n.stream())
        return dz1.to(ctx.in_dtypes[0]), dz2.to(ctx.in_dtypes[1]), None, None, None, None, None, None, None


def supcon_loss(z1, z2, labels=None, mask=None, *, t=0.07, sp_mode=SP_NONE, gamma=1e6, correct_grad=False,
                state: SupConState = None):
    """loss (0-dim tensor with grad_fn).  ``state`` receives the device-side statistics (rho = state.out[1])."""
    if state is None:
        state = SupConState()
    return _SupConFn.apply(z1, z2, labels, mask, float(t), int(sp_mode), float(gamma), bool(correct_grad), state)


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
        pooled = torch.empty(N, C, dtype=torch.float32, device=dev)
        pre = torch.empty(N, hid, dtype=torch.float32, device=dev) if mlp else None
        o = torch.empty(N, out_dim, dtype=torch.float32, device=dev)
        z = torch.empty(N, out_dim, dtype=torch.float32, device=dev)
        _n.call("spcl_proj_forward", _n.ptr(x), _n.dtype_code(x.dtype), N, H * W, C, cs, _n.ptr(w1c), _n.ptr(b1c),
                _n.ptr(w2c), _n.ptr(b2c), hid, out_dim, int(bool(normalize)), _n.ptr(pooled), _n.ptr(pre), _n.ptr(o),
                _n.ptr(z), _n.stream())
        ctx.save_for_backward(w1c, w2c, pooled, pre, o)
        ctx.meta = (N, H, W, C, cs, hid, out_dim, bool(normalize), x.dtype, feat.dtype)
        return z

    @staticmethod
    def backward(ctx, dz):
        w1c, w2c, pooled, pre, o = ctx.saved_tensors
        N, H, W, C, cs, hid, out_dim, normalize, xdt, fdt = ctx.meta
        dev = dz.device
        mlp = hid > 0
        dzc = dz.detach().contiguous().float()
        dw1 = torch.empty_like(w1c)
        db1 = torch.empty(w1c.shape[0], dtype=torch.float32, device=dev)
        dw2 = torch.empty_like(w2c) if mlp else None
        db2 = torch.empty(out_dim, dtype=torch.float32, device=dev) if mlp else None
        scratch = torch.empty(N * (out_dim + hid + C), dtype=torch.float32, device=dev)
        need_dfeat = ctx.needs_input_grad[0]
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
