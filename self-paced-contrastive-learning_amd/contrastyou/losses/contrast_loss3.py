"""HIP-backed mirror of the reference module ``contrastyou/losses/contrast_loss3.py``.

Same public names and call contract (reference line numbers cited per symbol); the arithmetic runs in the fused
gfx950 kernels of csrc/supcon.hip through the C ABI -- the [2n,2n] similarity matrix is never materialised unless
one of the hook taps (``sim_exp``, ``sim_logits``, ``pos_mask``, ``neg_mask``, ``sp_mask``) is read.

Host synchronisation: the reference synchronises 4x per call (two ``allclose``, ``.item()``, ``isnan``).  Here a
call enqueues kernels only.  With ``sync_checks=True`` (default, reference-identical error behaviour) ONE readback
of the 8-float result block performs the unit-norm assertion and the NaN check; with ``sync_checks=False`` nothing
synchronises and ``check()`` / ``downgrade_ratio`` read the block when asked.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
from torch import Tensor, nn

from ... import functional as F_hip

__all__ = ["SupConLoss1", "SelfPacedSupConLoss", "is_normalized", "exp_sim_temperature", "supcon_heads"]


def _capturing() -> bool:
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def is_normalized(feature: Tensor, dim=1) -> bool:
    """contrast_loss3.py:20-22 (host-side helper; the fused kernel evaluates the same test on device)."""
    norms = feature.norm(dim=dim)
    return bool(torch.allclose(norms, torch.ones_like(norms)))


def exp_sim_temperature(proj_feat1: Tensor, proj_feat2: Tensor, t: float) -> Tuple[Tensor, Tensor]:
    """contrast_loss3.py:25-31: (exp(S-max S), S-max S) for S = cat(z1,z2) cat(z1,z2)^T / t, materialised."""
    st = F_hip.SupConState()
    F_hip.supcon_loss(proj_feat1, proj_feat2, None, None, t=t, state=st)
    taps = F_hip.supcon_materialize(st, want=("sim_logits", "sim_exp"))
    return taps["sim_exp"], taps["sim_logits"]


class _SupConBase(nn.Module):
    _TAPS = ("sim_exp", "sim_logits", "pos_mask", "neg_mask", "sp_mask")

    def __init__(self, temperature: float, sync_checks: bool = True):
        super().__init__()
        self._t = temperature
        self.sync_checks = sync_checks
        self._state: Optional[F_hip.SupConState] = None
        self._taps_cache = None
        self._host_out = None
        self._lag = None  # ``check_lagged``: two pinned copies of the result block with their events, which one is pending

    def __getstate__(self):  # (copy / pickle: the pinned buffers and events stay behind)
        d = dict(self.__dict__)
        d["_lag"] = None
        return d

    # ---- inputs -------------------------------------------------------------------------------------------
    @staticmethod
    def _prepare_targets(proj_feat1: Tensor, proj_feat2: Tensor, target, mask):
        batch_size = proj_feat1.size(0)
        dev = proj_feat2.device
        labels_t = mask_t = None
        if mask is not None:  # contrast_loss3.py:43-46 / :128-131
            assert mask.shape == torch.Size([batch_size, batch_size])
            mask_t = mask.to(device=dev, dtype=torch.float32).contiguous()
        elif target is not None:  # :48-54 / :133-139  (labels are compared as floats, like torch.Tensor(list))
            if isinstance(target, Tensor):
                labels_t = target.to(device=dev, dtype=torch.float32).contiguous()
            else:
                labels_t = torch.tensor(list(target), dtype=torch.float32, device=dev)
            assert labels_t.numel() == batch_size, (labels_t.shape, batch_size)
        return labels_t, mask_t

    def _run(self, proj_feat1, proj_feat2, labels_t, mask_t, sp_mode, gamma, correct_grad, normalize_inputs=False):
        assert proj_feat1.shape == proj_feat2.shape, (proj_feat1.shape, proj_feat2.shape)  # :63 / :155
        self._state = F_hip.SupConState()
        self._taps_cache = None
        self._host_out = None
        stacked = F_hip.stacked_halves(proj_feat1, proj_feat2)
        if stacked is not None:  # the two views are torch.chunk halves of one projection: skip the chunk / cat copies
            proj_feat1, proj_feat2 = stacked, None
        loss = F_hip.supcon_loss(proj_feat1, proj_feat2, labels_t, mask_t, t=self._t, sp_mode=sp_mode, gamma=gamma,
                                 correct_grad=correct_grad, state=self._state, normalize_inputs=normalize_inputs)
        if self.sync_checks and not _capturing():
            self.check()  # (a captured step is checked after its replay: _INFONCEEpochHook.after_replay)
        return loss

    # ---- deferred contract checks -------------------------------------------------------------------------
    def _out(self):
        if self._host_out is None:
            self._host_out = self._state.out.tolist()  # the single device->host readback
        return self._host_out

    @staticmethod
    def _check_block(out):
        assert out[3] <= 1e-8 + 1e-5, "features need to be normalized first"
        if math.isnan(out[0]):
            raise RuntimeError(torch.tensor(out[0]))

    def check(self):
        """Unit-norm assertion (contrast_loss3.py:62,154) and NaN guard (:107-108,203-204) of the last call."""
        self._check_block(self._out())

    # A step replayed from a hipGraph is checked WITHOUT draining the queue: its result block is copied to pinned host memory
    # behind the replay (asynchronously, with an event), and what is LOOKED AT is the previous step's copy, whose event fired
    # a step ago.  The reference's errors are all raised (same types, same messages), one step after the step that earned
    # them -- whose optimizer update, part of the same graph, had run by the time of a readback anyway -- or when the
    # epoch's hooks close (``flush_check``).  With a readback per step the host cannot prepare step k + 1 while the GPU runs
    # step k: 1.0 ms of kernels + 0.74 ms of host work per step instead of max(...) of them -- the reference's own
    # configuration ran at 16 k slices/s instead of 27 k (tools/diag/pretrain_trainer_epochs.py).
    def check_lagged(self):
        if self._state is None or not self._state.out.is_cuda:
            return self.check()
        lag = self._lag
        if lag is None:
            lag = self._lag = {"slots": [], "pending": None, "next": 0}
        if not lag["slots"]:
            lag["slots"] = [(torch.empty(self._state.out.numel(), dtype=self._state.out.dtype).pin_memory(),
                             torch.cuda.Event()) for _ in range(2)]
        host, ev = lag["slots"][lag["next"]]
        host.copy_(self._state.out.reshape(-1), non_blocking=True)
        ev.record()
        prev, lag["pending"] = lag["pending"], lag["next"]
        lag["next"] ^= 1
        if prev is not None:
            self._check_slot(prev)

    def _check_slot(self, idx):
        host, ev = self._lag["slots"][idx]
        ev.synchronize()
        self._check_block(host.tolist())

    def flush_check(self):
        """look at the copy ``check_lagged`` has pending (the hooks call it when they close: no error is lost)"""
        lag = self._lag
        if lag and lag["pending"] is not None:
            idx, lag["pending"] = lag["pending"], None
            self._check_slot(idx)

    # ---- hook taps (contrast_loss3.py:83-88,175-178,188) --------------------------------------------------
    def _tap(self, name):
        if self._state is None:
            raise AttributeError(name)
        if self._taps_cache is None:
            self._taps_cache = F_hip.supcon_materialize(self._state)
        return self._taps_cache[name]

    sim_exp = property(lambda self: self._tap("sim_exp"))
    sim_logits = property(lambda self: self._tap("sim_logits"))
    pos_mask = property(lambda self: self._tap("pos_mask"))
    neg_mask = property(lambda self: self._tap("neg_mask"))


class SupConLoss1(_SupConBase):
    """contrast_loss3.py:34-110."""

    def __init__(self, temperature=0.07, exclude_other_pos=False, sync_checks=True):
        super().__init__(temperature, sync_checks)
        self._exclude_pos = exclude_other_pos

    def forward(self, proj_feat1, proj_feat2, target=None, mask: Tensor = None, normalize_inputs=False, **kwargs):
        """``normalize_inputs=True`` (not in the reference): the inputs are the projector's rows BEFORE its
        ``F.normalize`` (``ProjectionHead(...)(x, normalize=False)``); the loss is that of the normalised rows, the
        normalisation and its backward run inside the loss launch (training sizes) instead of in two launches of their own."""
        labels_t, mask_t = self._prepare_targets(proj_feat1, proj_feat2, target, mask)
        if not self._exclude_pos:
            return self._run(proj_feat1, proj_feat2, labels_t, mask_t, F_hip.SP_NONE, 1e6, False, normalize_inputs)
        if normalize_inputs:
            proj_feat1, proj_feat2 = F_hip.l2norm_rows(proj_feat1), F_hip.l2norm_rows(proj_feat2)
        # :97-100 -- each positive against the row's negatives only (csrc/supcon_xpos.hip); the taps are those of the
        # plain loss on the same inputs (same masks / logits), evaluated without a graph
        assert proj_feat1.shape == proj_feat2.shape, (proj_feat1.shape, proj_feat2.shape)
        self._state = F_hip.SupConState()
        self._taps_cache = self._host_out = None
        with torch.no_grad():
            F_hip.supcon_loss(proj_feat1, proj_feat2, labels_t, mask_t, t=self._t, state=self._state)
        out = torch.empty(8, dtype=torch.float32, device=proj_feat1.device)
        loss = F_hip.supcon_loss_exclude_other_pos(proj_feat1, proj_feat2, labels_t, mask_t, t=self._t, out=out)
        self._state.out = out
        if self.sync_checks and not _capturing():
            self.check()
        return loss


class SelfPacedSupConLoss(_SupConBase):
    """contrast_loss3.py:113-222."""

    def __repr__(self):
        return f"{self.__class__.__name__} with T: {self._t}, method: {self._weight_update} gamma: {self.__gamma}"

    def __init__(self, temperature=0.07, weight_update="hard", correct_grad=False, sync_checks=True, **kwargs):
        super().__init__(temperature, sync_checks)
        self._weight_update = weight_update
        self.__gamma = 1e6
        self._correct_grad = correct_grad

    def forward(self, proj_feat1, proj_feat2, target=None, mask: Tensor = None, normalize_inputs=False, **kwargs):
        """``normalize_inputs``: see ``SupConLoss1.forward``"""
        labels_t, mask_t = self._prepare_targets(proj_feat1, proj_feat2, target, mask)
        mode = F_hip.SP_HARD if self._weight_update == "hard" else F_hip.SP_SOFT  # :209-213
        return self._run(proj_feat1, proj_feat2, labels_t, mask_t, mode, self.__gamma, self._correct_grad, normalize_inputs)

    sp_mask = property(lambda self: self._tap("sp_mask"))

    @property
    def downgrade_ratio(self) -> float:
        """rho: mean self-paced weight over the positive pairs (:189-191), as a python float (reads the device)."""
        return float(self._out()[1])

    @property
    def downgrade_ratio_tensor(self) -> Tensor:
        """rho as a 0-dim device tensor (no synchronisation)."""
        return self._state.out[1]

    def set_gamma(self, gamma):  # :216-218
        self.__gamma = float(gamma)

    @property
    def age_param(self):  # :220-222
        return self.__gamma


_HEADS_MAX_ROWS = 1024  # 2n from which csrc/supcon.hip switches to the large-batch schedule (one head per call there)


def _adjacent_rows(rows, n):
    """the [K, n] float32 tensor whose rows ARE the K given vectors when they lie back to back in one allocation, else None"""
    b = rows[0]._base
    if b is None or any(r._base is not b or not r.is_contiguous() or r.dtype != torch.float32 for r in rows):
        return None
    if any(r.data_ptr() != rows[0].data_ptr() + k * n * 4 for k, r in enumerate(rows)):
        return None
    if b.dtype not in (torch.float32, torch.uint8) or not b.is_contiguous() or b.dim() != 1:
        return None
    byte_off = rows[0].data_ptr() - b.data_ptr()
    if byte_off % 4 or (b.data_ptr() % 4) or (b.numel() * b.element_size()) % 4:
        return None
    fb = b if b.dtype == torch.float32 else b.view(torch.float32)
    return torch.as_strided(fb, (len(rows), n), (n, 1), byte_off // 4)


def supcon_heads(criteria, projections, targets, normalize_inputs=False):
    """The K losses ``criteria[k](*torch.chunk(projections[k], 2), target=targets[k])`` in the launches of ONE
    (``spcl_supcon_forward_heads``): the K meta-label hooks of ``semi_seg/hooks/creator.py:102-124`` on one feature, each
    with its own label vector and its own age parameter.  Every criterion ends up in exactly the state a call of its own
    would leave (result block, workspace for the taps, ``downgrade_ratio``).  Returns the list of K 0-dim losses, or
    ``None`` when the group is not batchable: mixed classes / temperatures / weight rules, ``exclude_other_pos``,
    fewer than 2 or more than 4 heads, different shapes, or the large-batch size -- the caller then calls one by one."""
    K = len(criteria)
    if not 2 <= K <= 4 or len(projections) != K or len(targets) != K:
        return None
    c0 = criteria[0]
    if any(type(c) is not type(c0) or c._t != c0._t for c in criteria):
        return None
    if isinstance(c0, SelfPacedSupConLoss):
        if any(c._weight_update != c0._weight_update or c._correct_grad != c0._correct_grad for c in criteria):
            return None
        mode = F_hip.SP_HARD if c0._weight_update == "hard" else F_hip.SP_SOFT
        gammas, correct = [c.age_param for c in criteria], c0._correct_grad
    elif type(c0) is SupConLoss1:
        if any(c._exclude_pos for c in criteria):
            return None
        mode, gammas, correct = F_hip.SP_NONE, [1e6] * K, False
    else:
        return None
    z0 = projections[0]
    if (z0.dim() != 2 or z0.shape[0] % 2 or z0.shape[0] >= _HEADS_MAX_ROWS or z0.shape[1] > 4096
            or any(z.shape != z0.shape for z in projections)):
        return None
    n = z0.shape[0] // 2
    labels = []
    for tgt in targets:
        if tgt is None:
            return None
        lt = tgt if isinstance(tgt, Tensor) else torch.tensor(list(tgt), dtype=torch.float32, device=z0.device)
        lt = lt.to(device=z0.device, dtype=torch.float32)
        assert lt.numel() == n, (lt.shape, n)  # contrast_loss3.py:48-54 / :133-139
        labels.append(lt.reshape(n))
    labels_t = _adjacent_rows(labels, n)  # (the K hooks' slots of the epocher's stage sit back to back: no stacking launch)
    if labels_t is None:
        labels_t = torch.stack(labels)
    states = [F_hip.SupConState() for _ in range(K)]
    losses = F_hip.supcon_loss_heads(list(projections), labels_t, t=c0._t, sp_mode=mode, gammas=gammas,
                                     correct_grad=correct, states=states, normalize_inputs=normalize_inputs)
    for c, st in zip(criteria, states):
        c._state, c._taps_cache, c._host_out = st, None, None
    for c in criteria:
        if c.sync_checks and not _capturing():
            c.check()
    return list(losses)
