"""Fused RAdam over one flat fp32 parameter (``ddp.FlatParams.param``): the optimizer step of the pre-train
iteration as ONE HIP streaming kernel (+ a one-thread coefficient kernel) instead of torch's ~40 foreach launches.

Semantics are ``torch.optim.RAdam(lr, betas, eps, weight_decay, decoupled_weight_decay=False)`` (the reference builds
its RAdam from the un-vendored deepclustering2, ``contrastyou/trainer/base.py:62``; SURVEY.md section 8c fixes torch's as
the restatement).  The step counter and the learning rate live on the device, so a captured hipGraph replays
correctly; ``param_groups[i]["lr"]`` stays an ordinary float that LR schedulers may rewrite: it is pushed to the
device at every eager ``step()`` and by ``sync_lr()`` (call that between graph replays after a scheduler step)."""
from __future__ import annotations

import torch

from . import native as _n


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
    def step(self, closure=None, scalar_adds=None, grad_scale: float = 1.0):
        """``scalar_adds``: what ``contrastyou.meters.take_batch()`` returned -- the step's meter updates, performed by
        the first parameter's coefficient launch (one launch less per step).  ``grad_scale``: the update uses
        ``grad_scale * p.grad`` (ddp.FlatParams hands over the ranks' gradient SUM and 1 / world: the mean costs no pass
        of its own); 1.0 is the exact identity."""
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
