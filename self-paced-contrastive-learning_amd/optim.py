"""Fused RAdam over one flat fp32 parameter (``ddp.FlatParams.param``): the optimizer step of the pre-train
iteration as ONE HIP streaming kernel (+ a one-thread coefficient kernel, or -- in a staged step, whose host-written bytes
travel to the device anyway -- four coefficients computed on the host) instead of torch's ~40 foreach launches.

Semantics are ``torch.optim.RAdam(lr, betas, eps, weight_decay, decoupled_weight_decay=False)`` (the reference builds
its RAdam from the un-vendored deepclustering2, ``contrastyou/trainer/base.py:62``; SURVEY.md section 8c fixes torch's as
the restatement).  The step counter and the learning rate live on the device, so a captured hipGraph replays
correctly; ``param_groups[i]["lr"]`` stays an ordinary float that LR schedulers may rewrite: it is pushed to the
device at every eager ``step()`` and by ``sync_lr()`` (call that between graph replays after a scheduler step)."""
from __future__ import annotations

import math
import os

import numpy as np
import torch

from . import native as _n

_STAGED_COEF = os.environ.get("SPCL_RADAM_STAGED", "1") != "0"  # A/B switch: 0 keeps the coefficient launch in staged steps


def _ipow(b: float, e: int) -> float:
    """b ** e by repeated squaring, multiplication by multiplication what csrc/optim.hip ipow does (same bits)"""
    r = 1.0
    while e > 0:
        if e & 1:
            r *= b
        b *= b
        e >>= 1
    return r


def radam_coefficients(t: int, lr: float, beta1: float, beta2: float):
    """the step's three scalars (csrc/optim.hip radam_tick_kernel, torch.optim.RAdam's formulas, in double) + t; every
    operation is an IEEE double operation in the kernel's order: the floats are the kernel's, bit for bit"""
    lr = float(np.float32(lr))  # (the kernel reads the learning rate from a float32 device scalar)
    b1t, b2t = _ipow(beta1, t), _ipow(beta2, t)
    bc1, bc2 = 1.0 - b1t, 1.0 - b2t
    rho_inf = 2.0 / (1.0 - beta2) - 1.0
    rho_t = rho_inf - 2.0 * t * b2t / bc2
    if rho_t > 5.0:
        rect = math.sqrt((rho_t - 4.0) * (rho_t - 2.0) * rho_inf / ((rho_inf - 4.0) * (rho_inf - 2.0) * rho_t))
        return [lr / bc1, rect * math.sqrt(bc2), 1.0, float(t)]
    return [lr / bc1, 0.0, 0.0, float(t)]


class FusedRAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1):
            raise ValueError("invalid RAdam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        for group in self.param_groups:
            for p in group["params"]:
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise TypeError("FusedRAdam takes contiguous fp32 parameters (use ddp.FlatParams)")
                _n.require_gpu(p)
        self._step_host = {}  # id(p) -> host mirror of the device step counter while staged steps run (absent: unknown)

    def _state(self, p, group):
        st = self.state[p]
        if not st:
            st["step"] = torch.zeros((), dtype=torch.int64, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["lr_dev"] = torch.full((), float(group["lr"]), dtype=torch.float32, device=p.device)
            st["lr_host"] = float(group["lr"])
            st["coef"] = torch.zeros(4, dtype=torch.float32, device=p.device)
        return st

    def _staged_coef(self, p, group):
        """fill function of this parameter's stage slot: called once per staged step (stepgraph.StepStage.begin / bind)"""
        if id(p) not in self._step_host:
            self._step_host[id(p)] = int(self.state[p]["step"].item())  # (one readback: the first staged step, or after eager steps)
        t = self._step_host[id(p)] = self._step_host[id(p)] + 1
        if t >= 1 << 24:
            raise OverflowError("FusedRAdam: staged step count beyond 2^24 (the count travels as a float)")
        b1, b2 = group["betas"]
        return radam_coefficients(t, float(group["lr"]), float(b1), float(b2))

    def forget_staged_steps(self):
        """drop the host mirrors of the step counts: a staged step whose update launch did not happen left them one ahead of
        the device counters (``_staged_coef`` advances at fill time); the next staged step reads the device's counts back"""
        self._step_host.clear()

    @torch.no_grad()
    def sync_lr(self):
        """push the groups' current learning rates to the device (outside graph capture)."""
        for group in self.param_groups:
            for p in group["params"]:
                st = self._state(p, group)
                if st["lr_host"] != float(group["lr"]):
                    st["lr_dev"].fill_(float(group["lr"]))
                    st["lr_host"] = float(group["lr"])

    @torch.no_grad()
    def step(self, closure=None, scalar_adds=None, grad_scale: float = 1.0, stage=None):
        """``scalar_adds``: what ``contrastyou.meters.take_batch()`` returned -- the step's meter updates, performed by
        the first parameter's coefficient launch (one launch less per step).  ``grad_scale``: the update uses
        ``grad_scale * p.grad`` (ddp.FlatParams hands over the ranks' gradient SUM and 1 / world: the mean costs no pass
        of its own); 1.0 is the exact identity.  ``stage``: the epocher's ``stepgraph.StepStage`` while a staged step
        runs -- the step's scalar coefficients are then computed on the host from ``param_groups[i]["lr"]`` and a host
        mirror of the step count, and ride in the stage's upload (no coefficient launch, no ``sync_lr``)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        capturing = torch.cuda.is_current_stream_capturing()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                if g.dtype != torch.float32 or not g.is_contiguous():
                    raise TypeError("FusedRAdam needs a contiguous fp32 gradient")
                st = self._state(p, group)
                if not capturing and st["lr_host"] != float(group["lr"]):  # None after load_state_dict
                    st["lr_dev"].fill_(float(group["lr"]))
                    st["lr_host"] = float(group["lr"])
                k, src, dst, cnt = 0, None, None, None
                if scalar_adds is not None:
                    src, dst, cnt, k = scalar_adds[:4]
                    scalar_adds = None
                if _STAGED_COEF and stage is not None and stage.active:
                    coef = stage.bind(("radam", id(p)), 4, "f32", lambda b, p=p, group=group: self._staged_coef(p, group))
                    _n.call("spcl_radam_apply_staged", _n.ptr(p), _n.ptr(g), float(grad_scale), _n.ptr(st["exp_avg"]),
                            _n.ptr(st["exp_avg_sq"]), p.numel(), _n.ptr(st["step"]), _n.ptr(coef), float(b1), float(b2),
                            float(group["eps"]), float(group["weight_decay"]), k, src, dst, cnt, _n.stream())
                    continue
                self._step_host.pop(id(p), None)  # (the device counter advances by itself below)
                _n.call("spcl_radam_step_scaled", _n.ptr(p), _n.ptr(g), float(grad_scale), _n.ptr(st["exp_avg"]),
                        _n.ptr(st["exp_avg_sq"]), p.numel(), _n.ptr(st["step"]), _n.ptr(st["lr_dev"]), float(b1), float(b2), float(group["eps"]),
                        float(group["weight_decay"]), _n.ptr(st["coef"]), k, src, dst, cnt, _n.stream())
        if scalar_adds is not None:  # no parameter was stepped: the adds still have to happen
            _n.call("spcl_accumulate_scalars", scalar_adds[3], scalar_adds[0], scalar_adds[1], scalar_adds[2], _n.stream())
        return loss

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for group in self.param_groups:  # torch casts floating state to the parameter's dtype/device; restore ours
            for p in group["params"]:
                st = self.state.get(p)
                if st:
                    st["step"] = st["step"].to(device=p.device, dtype=torch.int64)
                    for k in ("lr_dev", "coef", "exp_avg", "exp_avg_sq"):
                        st[k] = st[k].to(device=p.device, dtype=torch.float32)
                    st["lr_host"] = None  # force a push of the group's lr at the next eager step
        self._step_host = {}
