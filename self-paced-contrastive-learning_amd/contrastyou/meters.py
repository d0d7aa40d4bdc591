"""Minimal meters for the hot path (the reference's ``contrastyou/meters`` package is out of scope): the
three meter names the InfoNCE hooks write (``loss``, ``sp_weight``, ``age_param``) plus ``reg_loss``/``lr``.
Values may be device tensors: they are accumulated ON DEVICE and read back once, in ``summary()`` -- the reference's
per-step ``.item()`` synchronisations (K19) disappear."""
from collections import OrderedDict
from contextlib import contextmanager

import torch


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
            self._dev[0].add_(value.detach().float().reshape(()), alpha=n)
            self._dev[1].add_(n)
        elif isinstance(value, (list, tuple)):
            for x in value:
                self.add(x, n)
        else:
            self._sum += float(value) * n
            self._n += n

    def summary(self):
        total, count = self._sum, self._n
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
