"""Mirror of the fine-tune trainer: ``FineTuneTrainer`` (semi_seg/trainers/new_trainer.py:17-64 -> ``SemiTrainer`` with
``activate_hooks = False`` and ``FineTuneEpocher``) on the loop of ``Trainer._start_training``
(contrastyou/trainer/base.py:94-121): per epoch one training epocher, evaluation on the validation and test loaders,
best / last checkpoints keyed on the validation ``DSC_mean``, scheduler step.  Optimizer / schedule as in the pre-train
mirror (flat parameter + fused RAdam, linear warm-up then cosine)."""
import os
from typing import Iterable, Optional

import torch
from torch import nn

from ... import ddp as _ddp
from ...optim import FusedRAdam
from ..epochers.finetune import EvalEpocher, FineTuneEpocher
from .pretrain import WarmupCosine


class FineTuneTrainer:
    activate_hooks = False

    def __init__(self, *, model: nn.Module, labeled_loader: Iterable, val_loader: Iterable, test_loader: Iterable = None,
                 unlabeled_loader: Iterable = None, criterion, save_dir: Optional[str] = None, max_epoch: int = 100,
                 num_batches: int = 100, device="cuda", lr=1e-7, weight_decay=1e-5, warmup_max=10, multiplier=400,
                 config=None, **kwargs):
        self._model = model
        self._labeled_loader, self._unlabeled_loader = labeled_loader, unlabeled_loader
        self._val_loader, self._test_loader = val_loader, test_loader
        self._criterion = criterion
        self._save_dir, self._max_epoch, self._num_batches, self._device = save_dir, max_epoch, num_batches, device
        self._optim_cfg = dict(lr=lr, weight_decay=weight_decay)
        self._sched_cfg = dict(warmup_max=warmup_max, multiplier=multiplier)
        self._config = config
        self._cur_epoch, self._start_epoch, self._best_score = 0, 0, -1.0
        self._optimizer = self._scheduler = self._flat = None
        self.history = []

    def init(self):
        self._model.to(self._device)
        _ddp.broadcast_state(self._model)
        self._flat = _ddp.FlatParams([p for p in self._model.parameters() if p.requires_grad])
        self._optimizer = FusedRAdam([self._flat.param], **self._optim_cfg)
        self._scheduler = WarmupCosine(self._optimizer, max_epoch=self._max_epoch, **self._sched_cfg)

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
        if self._optimizer is None:
            raise RuntimeError(f"{self.__class__.__name__} should call `init()` first")
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
            self._scheduler.step()
        return self.history

    def state_dict(self):
        return {"_model": self._model.state_dict(), "_optimizer": self._optimizer.state_dict(),
                "_scheduler": self._scheduler.state_dict(),
                "_buffers": {"_cur_epoch": self._cur_epoch, "_start_epoch": self._start_epoch,
                             "_best_score": self._best_score}}

    def load_state_dict(self, sd):
        self._model.load_state_dict(sd["_model"])
        self._optimizer.load_state_dict(sd["_optimizer"])
        self._scheduler.load_state_dict(sd["_scheduler"])
        for k, v in sd["_buffers"].items():
            setattr(self, k, v)

    def save_to(self, name):
        os.makedirs(self._save_dir, exist_ok=True)
        tmp = os.path.join(self._save_dir, name + ".tmp")
        torch.save(self.state_dict(), tmp)
        os.replace(tmp, os.path.join(self._save_dir, name))
