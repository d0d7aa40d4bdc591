"""Mirror of the fine-tune trainer: ``FineTuneTrainer`` (semi_seg/trainers/new_trainer.py:17-64 -> ``SemiTrainer`` with
``activate_hooks = False`` and ``FineTuneEpocher``) on the loop of ``Trainer._start_training``
(contrastyou/trainer/base.py:94-121): per epoch one training epocher, evaluation on the validation and test loaders,
best / last checkpoints keyed on the validation ``DSC_mean``, scheduler step.  Optimizer and schedule are built in
``init()`` from ``config["Optim"]`` / ``config["Scheduler"]`` as ``Trainer._init_optimizer`` / ``_init_scheduler`` do
(contrastyou/trainer/base.py:60-83; ``val.py:57-60`` constructs the trainer with ``config=global_config,
**config["Trainer"]`` and no learning rate of its own) -- flat parameter + fused RAdam, linear warm-up then cosine, as in
the pre-train mirror."""
import os
from typing import Iterable, Optional

import torch
from torch import nn

from ... import ddp as _ddp
from ...optim import FusedRAdam
from ..epochers.finetune import EvalEpocher, FineTuneEpocher
from .pretrain import WarmupCosine, build_optimizer, read_optim_sched


class FineTuneTrainer:
    activate_hooks = False

    def __init__(self, *, model: nn.Module, labeled_loader: Iterable, val_loader: Iterable, test_loader: Iterable = None,
                 unlabeled_loader: Iterable = None, criterion, save_dir: Optional[str] = None, max_epoch: int = 100,
                 num_batches: int = 100, device="cuda", lr=None, weight_decay=None, warmup_max=None, multiplier=None,
                 disable_bn: bool = False, two_stage: bool = False, config=None, **kwargs):
        """the reference's keyword set (``SemiTrainer.__init__``, semi_seg/trainers/new_trainer.py:20-35; ``name`` and other
        unknown ``config["Trainer"]`` keys fall into ``**kwargs`` there too) + the mirror's own optional overrides ``lr=``,
        ``weight_decay=``, ``warmup_max=``, ``multiplier=`` (given: they replace the config's value; a call without ``config``
        gets base.yaml's fine-tune values: lr 1e-7, weight decay 1e-5, warm-up 10 epochs to 300 x, config/base.yaml:11-18)"""
        self._model = self._inference_model = model
        self._labeled_loader, self._unlabeled_loader = labeled_loader, unlabeled_loader
        self._val_loader, self._test_loader = val_loader, test_loader
        self._criterion = self._sup_criterion = criterion
        self._disable_bn, self._two_stage = disable_bn, two_stage
        self._save_dir, self._max_epoch, self._num_batches, self._device = save_dir, max_epoch, num_batches, device
        self._optim_name, self._optim_cfg, self._sched_cfg = read_optim_sched(
            config, lr=lr, weight_decay=weight_decay, warmup_max=warmup_max, multiplier=multiplier, default_lr=1e-7,
            default_multiplier=300)
        self._config = config
        self._cur_epoch, self._start_epoch, self._best_score = 0, 0, 0  # trainer/_io.py:49-52
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

    def init(self):
        self._model.to(self._device)
        _ddp.broadcast_state(self._model)
        self._flat = _ddp.FlatParams([p for p in self._model.parameters() if p.requires_grad])
        self._optimizer = build_optimizer(self._optim_name, self._flat.param, self._optim_cfg)
        self._scheduler = None
        if self._sched_cfg is not None:  # trainer/base.py:72-73: no ``Scheduler`` section, no scheduler
            self._scheduler = WarmupCosine(self._optimizer, max_epoch=self._max_epoch, **self._sched_cfg)
        self.__initialized__ = True

    @property
    def train_epocher(self):
        return FineTuneEpocher

    def _create_tra_epoch(self):
        epocher = self.train_epocher(model=self._model, optimizer=self._optimizer, labeled_loader=self._labeled_loader,
                                     sup_criterion=self._criterion, num_batches=self._num_batches,
                                     cur_epoch=self._cur_epoch, device=self._device, flat_params=self._flat)
        epocher.init()
        return epocher

    def _create_eval_epoch(self, *, model, loader):
        epocher = EvalEpocher(model=model, loader=loader, sup_criterion=self._criterion, cur_epoch=self._cur_epoch,
                              device=self._device)
        epocher.init()
        return epocher

    def run_eval_epoch(self, *, model, loader):
        epocher = self._create_eval_epoch(model=model, loader=loader)
        stats = epocher.run()
        return stats, epocher.get_score()

    def start_training(self):
        if not self.__initialized__:
            raise RuntimeError(f"{self.__class__.__name__} should call `init()` first")
        from ... import stepgraph as _sg
        try:
            for self._cur_epoch in range(max(self._cur_epoch + 1, self._start_epoch), self._max_epoch + 1):
                train_metrics = self._create_tra_epoch().run()
                eval_metrics = test_metrics = None
                cur_score = float("nan")
                if _ddp.on_master():
                    eval_metrics, cur_score = self.run_eval_epoch(model=self._model, loader=self._val_loader)
                    if self._test_loader is not None:
                        test_metrics, _ = self.run_eval_epoch(model=self._model, loader=self._test_loader)
                best = self._best_score < cur_score
                if best:
                    self._best_score = cur_score
                if _ddp.on_master() and self._save_dir:
                    if best:
                        self.save_to("best.pth")
                    self.save_to("last.pth")
                self.history.append({"epoch": self._cur_epoch, "tra": train_metrics, "val": eval_metrics,
                                     "test": test_metrics, "score": cur_score})
                if self._scheduler is not None:
                    self._scheduler.step()
        finally:
            _sg.gc_release()  # (the epochers' captures keep the collector's heap frozen from one epoch to the next)
        return self.history

    def state_dict(self):
        return {"_model": self._model.state_dict(), "_optimizer": self._optimizer.state_dict(),
                "_scheduler": self._scheduler.state_dict() if self._scheduler is not None else None,
                "_buffers": {"_cur_epoch": self._cur_epoch, "_start_epoch": self._start_epoch,
                             "_best_score": self._best_score}}

    def load_state_dict(self, sd):
        self._model.load_state_dict(sd["_model"])
        self._optimizer.load_state_dict(sd["_optimizer"])
        if self._scheduler is not None and sd.get("_scheduler") is not None:
            self._scheduler.load_state_dict(sd["_scheduler"])
        for k, v in sd["_buffers"].items():
            setattr(self, k, v)

    def save_to(self, name):
        os.makedirs(self._save_dir, exist_ok=True)
        tmp = os.path.join(self._save_dir, name + ".tmp")
        torch.save(self.state_dict(), tmp)
        os.replace(tmp, os.path.join(self._save_dir, name))
