"""Minimum trainer for the encoder pre-train path: ``Trainer`` (contrastyou/trainer/base.py:23-155: hook registry,
optimizer + warm-up/cosine schedule, epoch loop), ``_PretrainTrainerMixin`` (semi_seg/trainers/new_pretrain.py:18-104)
and the checkpoint layout of ``contrastyou/trainer/_io.py:49-71`` (``_model``/``_optimizer``/``_scheduler``/
``__hooks__`` sub-dicts, so ``extract_model_state_dict`` style reuse of ``["_model"]`` interchanges).

The reference's RAdam / GradualWarmupScheduler come from the un-vendored ``deepclustering2``; restated by contract with
``torch.optim.RAdam`` (as the fused HIP kernel ``optim.FusedRAdam``) and a linear warm-up to ``multiplier`` x lr over ``warmup_max`` epochs followed by
CosineAnnealingLR(T_max=max_epoch-warmup_max, eta_min=1e-7)."""
import math
import os
from typing import Iterable, Optional

import torch
from torch import nn

from ... import ddp as _ddp
from ...optim import FusedRAdam
from ..epochers.pretrain import PretrainEncoderEpocher
from ..hooks.creator import feature_until_from_hooks


class WarmupCosine:
    def __init__(self, optimizer, *, max_epoch, warmup_max=10, multiplier=300, eta_min=1e-7):
        self.opt = optimizer
        self.base = [g["lr"] for g in optimizer.param_groups]
        self.max_epoch, self.warmup_max, self.multiplier, self.eta_min = max_epoch, warmup_max, multiplier, eta_min
        self.epoch = 0
        self._apply()

    def _lr(self, base):
        e = self.epoch
        if e <= self.warmup_max:
            return base * ((self.multiplier - 1.0) * e / max(1, self.warmup_max) + 1.0)
        top, t_max = base * self.multiplier, max(1, self.max_epoch - self.warmup_max)
        return self.eta_min + (top - self.eta_min) * (1 + math.cos(math.pi * (e - self.warmup_max) / t_max)) / 2

    def _apply(self):
        for g, b in zip(self.opt.param_groups, self.base):
            g["lr"] = self._lr(b)

    def step(self):
        self.epoch += 1
        self._apply()

    def state_dict(self):
        return {"epoch": self.epoch}

    def load_state_dict(self, sd):
        self.epoch = sd["epoch"]
        self._apply()


class PretrainEncoderTrainer:
    def __init__(self, *, model: nn.Module, chain_dataloader: Iterable, save_dir: Optional[str] = None,
                 max_epoch: int = 80, num_batches: int = 200, device="cuda", lr=5e-7, weight_decay=1e-5,
                 warmup_max=10, multiplier=400, **kwargs):
        self._model = model
        self._chain_dataloader = chain_dataloader
        self._save_dir = save_dir
        self._max_epoch, self._num_batches, self._device = max_epoch, num_batches, device
        self._optim_cfg = dict(lr=lr, weight_decay=weight_decay)
        self._sched_cfg = dict(warmup_max=warmup_max, multiplier=multiplier)
        self.__hooks__ = nn.ModuleList()
        self.forward_until = None
        self._cur_epoch, self._start_epoch = 0, 0
        self._optimizer = self._scheduler = self._flat = None
        self.history = []

    # trainer/base.py:49-58
    def register_hooks(self, *hooks):
        assert self._optimizer is None, "`register_hook` must be called before `init()`"
        for h in hooks:
            self.__hooks__.append(h)
        self.forward_until = feature_until_from_hooks(*hooks)

    # trainer/base.py:44-47,60-83
    def init(self):
        self._model.to(self._device)
        self.__hooks__.to(self._device)
        _ddp.broadcast_state(self._model, self.__hooks__)
        params = [p for p in self._model.parameters() if p.requires_grad]
        hook_params = [p for h in self.__hooks__ for p in h.parameters()]
        # the reference gives model and hook parameters two groups with identical lr / weight decay
        # (trainer/base.py:62-68): one flat parameter is the same optimisation problem
        self._flat = _ddp.FlatParams(params + hook_params)
        self._optimizer = FusedRAdam([self._flat.param], **self._optim_cfg)  # torch.optim.RAdam semantics, one launch
        self._scheduler = WarmupCosine(self._optimizer, max_epoch=self._max_epoch, **self._sched_cfg)

    def _create_tra_epoch(self):
        epocher = PretrainEncoderEpocher(model=self._model, optimizer=self._optimizer,
                                         chain_dataloader=self._chain_dataloader, num_batches=self._num_batches,
                                         cur_epoch=self._cur_epoch, device=self._device,
                                         inference_until=self.forward_until or "Conv5", flat_params=self._flat)
        epocher.add_hooks([h() for h in self.__hooks__])
        epocher.init()
        return epocher

    # trainers/new_pretrain.py:69-85
    def start_training(self):
        for self._cur_epoch in range(max(self._cur_epoch + 1, self._start_epoch) if self._cur_epoch else 0,
                                     self._max_epoch):
            stats = self._create_tra_epoch().run()
            self.history.append(stats)
            self._scheduler.step()
            if self._save_dir and _ddp.on_master():
                self.save_to("last.pth")
        return self.history

    # trainer/_io.py:49-71,120-134
    def state_dict(self):
        return {"_model": self._model.state_dict(), "_optimizer": self._optimizer.state_dict(),
                "_scheduler": self._scheduler.state_dict(), "__hooks__": self.__hooks__.state_dict(),
                "_hook_schedulers": [getattr(s, "_scheduler").state_dict() if hasattr(s, "_scheduler") else None
                                     for h in self.__hooks__ for s in getattr(h, "_hooks", [h])],
                "_buffers": {"_cur_epoch": self._cur_epoch, "_start_epoch": self._start_epoch}}

    def load_state_dict(self, sd):
        self._model.load_state_dict(sd["_model"])
        self._optimizer.load_state_dict(sd["_optimizer"])
        self._scheduler.load_state_dict(sd["_scheduler"])
        self.__hooks__.load_state_dict(sd["__hooks__"])
        subs = [s for h in self.__hooks__ for s in getattr(h, "_hooks", [h])]
        for s, ssd in zip(subs, sd.get("_hook_schedulers", [])):
            if ssd is not None and hasattr(s, "_scheduler"):
                s._scheduler.load_state_dict(ssd)
        self._cur_epoch = sd["_buffers"]["_cur_epoch"]
        self._start_epoch = sd["_buffers"]["_start_epoch"]

    def save_to(self, name):
        os.makedirs(self._save_dir, exist_ok=True)
        tmp = os.path.join(self._save_dir, name + ".tmp")
        torch.save(self.state_dict(), tmp)
        os.replace(tmp, os.path.join(self._save_dir, name))

    def resume_from_path(self, path):
        self.load_state_dict(torch.load(path, map_location="cpu"))
