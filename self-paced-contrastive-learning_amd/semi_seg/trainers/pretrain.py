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
from ..epochers.pretrain import PretrainDecoderEpocher, PretrainEncoderEpocher
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


def read_optim_sched(config, *, lr=None, weight_decay=None, warmup_max=None, multiplier=None, default_lr=1e-7,
                     default_multiplier=400):
    """``Trainer._init_optimizer`` / ``_init_scheduler`` (contrastyou/trainer/base.py:60-83) as data: the optimizer's name
    and keyword set from ``config["Optim"]`` (every key but ``name`` / ``pre_lr`` / ``ft_lr`` reaches the optimizer, :64),
    the scheduler's from ``config["Scheduler"]`` (``None`` when the section is absent: no scheduler, :72-73).  The mirror's
    own keywords (``lr=`` ...) only override when given; the defaults only fill a section-less (``config=None``) call."""
    optim_cfg = dict((config or {}).get("Optim", {}))
    name = optim_cfg.pop("name", "RAdam")
    for k in ("pre_lr", "ft_lr"):
        optim_cfg.pop(k, None)
    if lr is not None:
        optim_cfg["lr"] = lr
    if weight_decay is not None:
        optim_cfg["weight_decay"] = weight_decay
    optim_cfg.setdefault("lr", default_lr)
    optim_cfg.setdefault("weight_decay", 1e-5)
    sched = (config or {}).get("Scheduler", None)
    if config is None or sched is not None or warmup_max is not None or multiplier is not None:
        sched = dict(sched or {})
        if warmup_max is not None:
            sched["warmup_max"] = warmup_max
        if multiplier is not None:
            sched["multiplier"] = multiplier
        sched.setdefault("warmup_max", 10)
        sched.setdefault("multiplier", default_multiplier)
    return name, optim_cfg, sched


def build_optimizer(name, flat_param, optim_cfg):
    if name == "RAdam":
        return FusedRAdam([flat_param], **optim_cfg)  # torch.optim.RAdam semantics, HIP kernel
    if hasattr(torch.optim, name):
        return getattr(torch.optim, name)([flat_param], **optim_cfg)
    raise KeyError(name)


class PretrainEncoderTrainer:
    """``PretrainEncoderTrainer`` as ``main_pretrain_encoder.py:54-72`` drives it.

    Accepts the reference's constructor call unchanged --
    ``PretrainEncoderTrainer(model=, labeled_loader=, unlabeled_loader=, val_loader=, test_loader=, criterion=,
    config=, save_dir=, **config["Trainer"])`` (``SemiTrainer.__init__`` ``semi_seg/trainers/new_trainer.py:20-35`` +
    ``_PretrainTrainerMixin.__init__`` ``new_pretrain.py:36-45``): optimiser and schedule are read from
    ``config["Optim"]`` / ``config["Scheduler"]`` in ``init()`` (``contrastyou/trainer/base.py:60-83``), the contrastive
    loader is built from ``unlabeled_loader`` and ``config["ContrastiveLoaderParams"]`` (a missing section raises
    ``RuntimeError`` as the reference does) -- or the mirror's own short form ``chain_dataloader=`` with keyword
    hyper-parameters.  Epoch bookkeeping follows ``new_pretrain.py:69-85``: ``range(max(_cur_epoch + 1, _start_epoch),
    max_epoch)``, i.e. a fresh trainer runs epochs 1 .. max_epoch - 1 and a resumed one continues after the saved epoch."""
    RUN_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))),
                            "runs2")

    def __init__(self, *, model: nn.Module, chain_dataloader: Optional[Iterable] = None, labeled_loader=None,
                 unlabeled_loader=None, val_loader=None, test_loader=None, criterion=None, config: Optional[dict] = None,
                 save_dir: Optional[str] = None, max_epoch: int = 80, num_batches: int = 200, device="cuda",
                 lr=None, weight_decay=None, warmup_max=None, multiplier=None, two_stage: bool = False,
                 disable_bn: bool = False, **kwargs):
        self._model = model
        self._labeled_loader, self._unlabeled_loader = labeled_loader, unlabeled_loader
        self._val_loader, self._test_loader = val_loader, test_loader
        self._criterion = self._sup_criterion = criterion
        self._two_stage, self._disable_bn = two_stage, disable_bn
        self._config = config
        self._save_dir = save_dir
        self._max_epoch, self._num_batches, self._device = max_epoch, num_batches, device
        self._optim_name, self._optim_cfg, self._sched_cfg = read_optim_sched(
            config, lr=lr, weight_decay=weight_decay, warmup_max=warmup_max, multiplier=multiplier, default_lr=5e-7)
        if chain_dataloader is None:
            if config is None or "ContrastiveLoaderParams" not in config:  # new_pretrain.py:38-40
                raise RuntimeError("`ContrastiveLoaderParams` should be found in config, given \n`" +
                                   ", ".join((config or {}).keys()) + "`")
            from ..data import get_contrastive_dataloader  # row N2
            chain_dataloader, self._monitor_loader = get_contrastive_dataloader(
                unlabeled_loader, config["ContrastiveLoaderParams"], device=device)
        self._chain_dataloader = self._contrastive_loader = chain_dataloader
        self.__hooks__ = nn.ModuleList()
        self._inference_until = None
        self._cur_epoch, self._start_epoch, self._best_score = 0, 0, 0
        self._optimizer = self._scheduler = self._flat = None
        self.__initialized__ = False
        self.history = []
        if save_dir and config is not None and _ddp.on_master():  # trainer/base.py:40-41 (dump_config)
            os.makedirs(save_dir, exist_ok=True)
            import yaml
            with open(os.path.join(save_dir, "config.yaml"), "w") as f:
                yaml.safe_dump(config, f)

    @property
    def save_dir(self):
        return str(self._save_dir)

    @staticmethod
    def on_master():
        return _ddp.on_master()

    # new_pretrain.py:47-62
    @property
    def forward_until(self):
        if self._inference_until is None:
            return list(type(self._model).decoder_names)[-1]
        return self._inference_until

    @forward_until.setter
    def forward_until(self, forward_until):
        if isinstance(forward_until, str):
            if forward_until == "all":
                self._inference_until = None
                return
            assert forward_until in type(self._model).arch_elements, forward_until
        self._inference_until = forward_until

    # trainer/base.py:49-58
    def register_hook(self, hook):
        from ...contrastyou.hooks.base import TrainerHook
        assert isinstance(hook, TrainerHook), hook
        self.__hooks__.append(hook)

    def register_hooks(self, *hooks):
        if self.__initialized__:
            raise RuntimeError("`register_hook must be called before `init()``")
        for h in hooks:
            self.register_hook(h)

    # trainer/base.py:44-47,60-83
    def init(self):
        self._model.to(self._device)
        self.__hooks__.to(self._device)
        _ddp.broadcast_state(self._model, self.__hooks__)
        # call inside ``model.set_grad(False, start=until, include_start=False)`` (main_pretrain_encoder.py:69): only
        # parameters that require grad join the flat parameter, exactly the reference's optimiser membership
        params = [p for p in self._model.parameters() if p.requires_grad]
        hook_params = [p for h in self.__hooks__ for p in h.parameters()]
        # the reference gives model and hook parameters two groups with identical hyper-parameters
        # (trainer/base.py:62-68): one flat parameter is the same optimisation problem
        self._flat = _ddp.FlatParams(params + hook_params)
        self._optimizer = build_optimizer(self._optim_name, self._flat.param, self._optim_cfg)
        self._scheduler = None
        if self._sched_cfg is not None:
            self._scheduler = WarmupCosine(self._optimizer, max_epoch=self._max_epoch, **self._sched_cfg)
        self.__initialized__ = True

    train_epocher = PretrainEncoderEpocher

    def _create_tra_epoch(self):
        epocher = self.train_epocher(model=self._model, optimizer=self._optimizer,
                                         chain_dataloader=self._chain_dataloader, num_batches=self._num_batches,
                                         cur_epoch=self._cur_epoch, device=self._device,
                                         inference_until=self._inference_until, flat_params=self._flat)
        epocher.add_hooks([h() for h in self.__hooks__])
        epocher.init()
        self._last_epocher = epocher
        return epocher

    def run_tra_epoch(self):
        return self._create_tra_epoch().run()

    # trainer/base.py:85-92, trainers/new_pretrain.py:69-85
    def start_training(self):
        if not self.__initialized__:
            raise RuntimeError(f"{self.__class__.__name__} should call `init()` first")
        from ... import stepgraph as _sg
        try:
            for self._cur_epoch in range(max(self._cur_epoch + 1, self._start_epoch), self._max_epoch):
                stats = self.run_tra_epoch()
                self.history.append(stats)
                if self._scheduler is not None:
                    self._scheduler.step()
                if self._save_dir and _ddp.on_master():
                    self.save_to("last.pth")
        finally:
            _sg.gc_release()  # (the epochers' captures keep the collector's heap frozen from one epoch to the next)
        return self.history

    # trainer/_io.py:49-71,120-134
    def state_dict(self):
        return {"_model": self._model.state_dict(), "_optimizer": self._optimizer.state_dict(),
                "_scheduler": self._scheduler.state_dict() if self._scheduler is not None else None,
                "__hooks__": self.__hooks__.state_dict(),
                "_hook_schedulers": [getattr(s, "_scheduler").state_dict() if hasattr(s, "_scheduler") else None
                                     for h in self.__hooks__ for s in getattr(h, "_hooks", [h])],
                "_buffers": {"_cur_epoch": self._cur_epoch, "_start_epoch": self._start_epoch,
                             "_best_score": self._best_score}}

    def load_state_dict(self, sd):
        self._model.load_state_dict(sd["_model"])
        self._optimizer.load_state_dict(sd["_optimizer"])
        if self._scheduler is not None and sd.get("_scheduler") is not None:
            self._scheduler.load_state_dict(sd["_scheduler"])
        self.__hooks__.load_state_dict(sd["__hooks__"])
        subs = [s for h in self.__hooks__ for s in getattr(h, "_hooks", [h])]
        for s, ssd in zip(subs, sd.get("_hook_schedulers", [])):
            if ssd is not None and hasattr(s, "_scheduler"):
                s._scheduler.load_state_dict(ssd)
        self._cur_epoch = sd["_buffers"]["_cur_epoch"]
        self._start_epoch = sd["_buffers"]["_start_epoch"]
        self._best_score = sd["_buffers"].get("_best_score", 0)

    def save_to(self, save_name=None, name=None):
        name = save_name or name
        os.makedirs(self._save_dir, exist_ok=True)
        tmp = os.path.join(self._save_dir, name + ".tmp")
        torch.save(self.state_dict(), tmp)
        os.replace(tmp, os.path.join(self._save_dir, name))

    def resume_from_path(self, path):
        self.load_state_dict(torch.load(path, map_location="cpu"))


class PretrainDecoderTrainer(PretrainEncoderTrainer):
    """``PretrainDecoderTrainer`` (new_pretrain.py:107-110), driven by ``main_pretrain_decoder.py:53-71`` with
    ``forward_until`` = a decoder feature and ``model.set_grad(True, start="Conv5", end=until, include_start=False)``
    inside ``model.set_grad(False)`` (row N3)."""
    train_epocher = PretrainDecoderEpocher
