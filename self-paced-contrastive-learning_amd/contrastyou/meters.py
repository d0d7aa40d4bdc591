"""Minimal meters for the hot path (the reference's ``contrastyou/meters`` package is out of scope): the
three meter names the InfoNCE hooks write (``loss``, ``sp_weight``, ``age_param``) plus ``reg_loss``/``lr``.
Values may be device tensors: they are accumulated ON DEVICE and read back once, in ``summary()`` -- the reference's
per-step ``.item()`` synchronisations (K19) disappear."""
from collections import OrderedDict
from contextlib import contextmanager

import torch


# ---- batched device accumulation: between begin_batch() and flush_batch() the device-scalar adds of ALL meters are
# remembered and then performed by ONE kernel launch (spcl_accumulate_scalars) instead of two tiny launches per add.
_BATCH = None


def begin_batch():
    """start remembering device adds.  Anything still pending from an earlier, unfinished batch (a step that raised, an
    aborted hipGraph capture whose tensors no longer exist) is dropped, not replayed."""
    global _BATCH
    _BATCH = []


def take_batch(limit=8):
    """hand the first ``limit`` remembered adds to a caller that performs them inside a launch of its own (the fused
    optimizer's coefficient kernel, optim.FusedRAdam.step(scalar_adds=...)); the rest stays queued for flush_batch().
    Returns (src, dst, count) ctypes arrays and their length, or None when nothing is pending."""
    global _BATCH
    if not _BATCH:
        return None
    from ctypes import c_float, c_void_p
    part, _BATCH = _BATCH[:limit], _BATCH[limit:]
    dsts = [d.data_ptr() for _, d, _ in part]
    if len(set(dsts)) != len(dsts):  # a meter added to twice in one step: keep the plain path's chunking
        _BATCH = part + _BATCH
        return None
    k = len(part)
    return ((c_void_p * k)(*[v.data_ptr() for v, _, _ in part]), (c_void_p * k)(*dsts),
            (c_float * k)(*[float(n) for _, _, n in part]), k, part)


def flush_batch():
    """perform the remembered adds (same arithmetic: sum += n * value, count += n) and leave batching mode"""
    global _BATCH
    pending, _BATCH = _BATCH, None
    if not pending:
        return
    from .. import native as _n
    from ctypes import c_float, c_void_p
    for i in range(0, len(pending), 8):
        part = pending[i:i + 8]
        k = len(part)
        src = (c_void_p * k)(*[v.data_ptr() for v, _, _ in part])
        dst = (c_void_p * k)(*[d.data_ptr() for _, d, _ in part])
        cnt = (c_float * k)(*[float(n) for _, _, n in part])
        _n.call("spcl_accumulate_scalars", k, src, dst, cnt, _n.stream())


# ---- host-side adds of a captured step: a hipGraph replay repeats the step's device work but not the python floats its
# hooks hand to the meters (``age_param``); between begin_host_log() and end_host_log() they are remembered so that
# stepgraph.StepGraph can re-apply them after every replay.
_HOST_LOG = None


def begin_host_log():
    global _HOST_LOG
    _HOST_LOG = []


def end_host_log():
    global _HOST_LOG
    log, _HOST_LOG = _HOST_LOG, None
    return log or []


class AverageValueMeter:
    """Running mean.  Tensor values are accumulated IN PLACE into persistent device scalars (sum and count), so the
    accumulation is a pair of tiny device ops that also replay correctly from a captured hipGraph."""

    def __init__(self):
        self.reset()

    def reset(self):
        self._sum = 0.0
        self._n = 0
        self._dev = None  # [sum, count] on the value's device

    def add(self, value, n=1):
        if isinstance(value, torch.Tensor):
            if self._dev is None:
                self._dev = torch.zeros(2, dtype=torch.float32, device=value.device)
            if (_BATCH is not None and value.is_cuda and value.dtype == torch.float32 and value.numel() == 1
                    and self._dev.device == value.device):
                _BATCH.append((value.detach(), self._dev, n))  # keeps the value alive until the flush
                return
            self._dev[0].add_(value.detach().float().reshape(()), alpha=n)
            self._dev[1].add_(n)
        elif isinstance(value, (list, tuple)):
            for x in value:
                self.add(x, n)
        else:
            if _HOST_LOG is not None:
                _HOST_LOG.append((self, float(value), n))
            self._sum += float(value) * n
            self._n += n

    def retract(self, value, n=1):
        """take back one host-side ``add(value, n)`` (a step whose capture failed is run again eagerly, stepgraph.py)"""
        self._sum -= float(value) * n
        self._n -= n

    def summary(self):
        total, count = self._sum, self._n
        if _BATCH:
            flush_batch()
            begin_batch()
        if self._dev is not None:
            s, c = self._dev.tolist()  # the one device->host readback
            total, count = total + s, count + c
        return {"mean": total / count if count else float("nan")}


class MeterInterface:
    """Groups of named meters; ``focus_on(group)`` scopes registration/lookup like the reference's interface."""

    def __init__(self, default_focus="tra"):
        self._groups = OrderedDict()
        self._focus = default_focus

    @contextmanager
    def focus_on(self, group):
        prev, self._focus = self._focus, group
        try:
            yield self
        finally:
            self._focus = prev

    def register_meter(self, name, meter):
        self._groups.setdefault(self._focus, OrderedDict())[name] = meter

    def delete_meters(self, names):
        for n in names:
            self._groups.get(self._focus, {}).pop(n, None)

    def __getitem__(self, name):
        return self._groups[self._focus][name]

    def __contains__(self, name):
        return name in self._groups.get(self._focus, {})

    def statistics(self):
        return {g: {k: m.summary() for k, m in ms.items()} for g, ms in self._groups.items()}

    def reset(self):
        for ms in self._groups.values():
            for m in ms.values():
                m.reset()


class UniversalDice:
    """Mirror of ``contrastyou/meters/general_dice_meter.py:19-175``: per-group Dice from accumulated per-sample
    intersection / union counts, ``DSC = (2 I + 1e-6) / (U + 1e-6)``, mean / std over groups, ``summary()`` with
    ``DSC{i}`` for the reported axes and ``DSC_mean``.

    ``add(pred, target, group_name)`` takes class-coded maps of equal shape (``logits.max(1)[1]`` and the label map, as
    the epochers call it, new_epocher.py:89,282).  The counts come from one HIP launch (``spcl_dice_counts``) and stay
    on the device; nothing is read back before ``value()`` / ``summary()``."""

    def __init__(self, C=4, report_axises=None) -> None:
        assert report_axises is None or isinstance(report_axises, (list, tuple)), \
            f"`report_axises` should be either None or an iterator, given {type(report_axises)}"
        if report_axises is not None:
            assert max(report_axises) <= C, "Incompatible parameter of `C`={} and `report_axises`={}".format(
                C, report_axises)
        self._C = C
        self._report_axis = list(range(self._C)) if report_axises is None else report_axises
        self.reset()

    def reset(self):
        self._intersections, self._unions, self._group_names = [], [], []
        self._n = 0

    def add(self, pred, target, group_name=None):
        from .. import functional as F_hip
        assert pred.shape == target.shape, \
            f"incompatible shape of `pred` and `target`, given {pred.shape} and {target.shape}."
        assert not pred.requires_grad and not target.requires_grad
        B = pred.shape[0]
        if group_name is not None and not isinstance(group_name, str):
            if isinstance(group_name, (list, tuple)):
                assert len(group_name) == B and isinstance(group_name[0], str)
            else:
                raise TypeError(f"type of `group_name` wrong {type(group_name)}")
        if pred.is_floating_point():
            raise NotImplementedError("UniversalDice mirror takes class-coded (integer) maps, as the epochers pass them")
        names = [str(self._n) + f"_{i:03d}" for i in range(B)]  # slice-based dice
        if group_name is not None:
            names = [group_name] * B if isinstance(group_name, str) else list(group_name)
        inter, union = F_hip.dice_counts(pred, target, self._C)
        self.add_counts(inter, union, names)

    def group_names_for(self, B, group_name=None):
        """the per-sample group names ``add`` would record for a batch of ``B`` samples"""
        if group_name is None:
            return [str(self._n) + f"_{i:03d}" for i in range(B)]
        return [group_name] * B if isinstance(group_name, str) else list(group_name)

    def add_counts(self, inter, union, names):
        """record one batch's per-sample intersection / union counts ([B, C] device tensors the caller hands over)"""
        assert inter.shape == union.shape and inter.shape[0] == len(names) and inter.shape[1] == self._C
        self._intersections.append(inter)
        self._unions.append(union)
        self._group_names.extend(names)
        self._n += 1

    @property
    def group_names(self):
        return sorted(set(self._group_names))

    @property
    def log(self):
        if self._n == 0:
            return None
        import numpy as np
        inter = torch.cat(self._intersections, 0).cpu().double()  # the one device->host readback
        union = torch.cat(self._unions, 0).cpu().double()
        arr = np.asarray(self._group_names)
        rows = []
        for name in self.group_names:
            idx = torch.from_numpy(arr == name)
            rows.append((2 * inter[idx].sum(0) + 1e-6) / (union[idx].sum(0) + 1e-6))
        return torch.stack(rows, 0).float()

    def value(self, **kwargs):
        if self._n == 0:
            return [float("nan")] * self._C, [float("nan")] * self._C
        d = self.log
        return d.mean(0), d.std(0) if d.shape[0] > 1 else torch.full((self._C,), float("nan"))

    def summary(self) -> dict:
        means, _ = self.value()
        report = {f"DSC{i}": float(means[i]) for i in self._report_axis}
        report["DSC_mean"] = sum(report.values()) / len(report) if report else float("nan")
        return report

    def __repr__(self):
        return f"C={self._C}, report_axis={self._report_axis}\n\t{self.summary()}"
