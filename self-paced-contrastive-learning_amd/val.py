"""The tail of ``main_pretrain_encoder.main`` (:35-38) on the HIP-backed mirror: once ``worker()`` has returned the
pre-trained model, fine-tune it for every labelled ratio from the SAME pre-trained weights and score the Dice.

Interface of the reference's repo-root ``val.py`` (``val`` :24-42, ``_val`` :45-66, ``switch_model_device`` :16-21): keyword-only
arguments of the same names and meaning.  What it has to do, in this file's own words:

* ``val``: keep a CPU snapshot of the weights; per ratio restore it and -- with python's, numpy's and torch's generators seeded
  to ``seed`` for the duration -- run one fine-tuning (``_val``);
* ``_val``: work on copies of the config sections, write the ratio into ``Data.labeled_scan_num`` (section and global config),
  build the four loaders with ``get_data(..., pretrain=False)``, train a ``FineTuneTrainer`` that takes optimizer and schedule
  from the config (no learning rate of its own) under ``<save_dir>/tra/num_labeled_scan_<n>``, mark the directory done.

``val`` returns the trainers (the reference returns None): their ``history`` / ``_best_score`` are what a caller wants to see."""
import copy
import os
import random
from contextlib import contextmanager
from typing import Any, Dict, List

import numpy as np
import torch
from torch import nn

from .contrastyou import success
from .contrastyou.losses.kl import KL_div
from .semi_seg.data.creator import get_data
from .semi_seg.trainers.finetune import FineTuneTrainer

_SECTIONS = (("data_params", "Data"), ("labeled_loader_params", "LabeledLoader"),
             ("unlabeled_loader_params", "UnlabeledLoader"), ("trainer_params", "Trainer"))


class switch_model_device:
    """``with switch_model_device(model, "cpu"): ...`` -- the model visits ``device`` and returns to where it was"""

    def __init__(self, model: nn.Module, device: str = "cpu"):
        self._model, self._target, self._home = model, device, None

    def __enter__(self):
        self._home = next(self._model.parameters()).device
        self._model.to(self._target)
        return self._model

    def __exit__(self, *exc):
        self._model.to(self._home)
        return False


@contextmanager
def fix_all_seed_within_context(seed):
    """every generator a data split or an augmentation may draw from, seeded inside the block and put back after it
    (contrastyou/utils/utils.py:156-173)"""
    saved = {"py": random.getstate(), "np": np.random.get_state(), "torch": torch.random.get_rng_state(),
             "cuda": torch.cuda.get_rng_state_all() if torch.cuda.is_available() else None}
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if saved["cuda"] is not None:
        torch.cuda.manual_seed_all(seed)
    try:
        yield
    finally:
        random.setstate(saved["py"])
        np.random.set_state(saved["np"])
        torch.random.set_rng_state(saved["torch"])
        if saved["cuda"] is not None:
            torch.cuda.set_rng_state_all(saved["cuda"])


def val(*, model: nn.Module, save_dir: str, base_config: Dict[str, Any], labeled_ratios: List[float], seed: int = 10):
    with switch_model_device(model, device="cpu") as on_cpu:
        snapshot = {name: t.detach().clone() for name, t in on_cpu.state_dict().items()}
    sections = {kw: base_config[key] for kw, key in _SECTIONS}
    finished = []
    for ratio in labeled_ratios:
        model.load_state_dict(snapshot)  # every ratio starts from the pre-trained weights
        with fix_all_seed_within_context(seed):
            finished.append(_val(model=model, labeled_data_ratio=ratio, main_save_dir=save_dir, global_config=base_config,
                                 **sections))
    return finished


def _ratio_dir(root: str, labeled_loader) -> str:
    n_scans = len(labeled_loader.dataset.get_scan_list())
    return os.path.join(root, "tra", f"num_labeled_scan_{n_scans}")


def _val(*, model: nn.Module, labeled_data_ratio: float, data_params: Dict[str, Any],
         labeled_loader_params: Dict[str, Any], unlabeled_loader_params: Dict[str, Any], main_save_dir: str,
         trainer_params: Dict[str, Any], global_config: Dict[str, Any]):
    ratio = float(labeled_data_ratio)
    data_cfg, trainer_cfg, config = copy.deepcopy(data_params), copy.deepcopy(trainer_params), copy.deepcopy(global_config)
    data_cfg["labeled_scan_num"] = ratio
    config["Data"]["labeled_scan_num"] = ratio
    names = ("labeled_loader", "unlabeled_loader", "val_loader", "test_loader")
    loaders = dict(zip(names, get_data(data_params=data_cfg, labeled_loader_params=labeled_loader_params,
                                       unlabeled_loader_params=unlabeled_loader_params, pretrain=False)))
    trainer_cfg["save_dir"] = _ratio_dir(main_save_dir, loaders["labeled_loader"])
    trainer = FineTuneTrainer(model=model, criterion=KL_div(verbose=False), config=config, **loaders, **trainer_cfg)
    trainer.init()
    trainer.start_training()
    success(save_dir=trainer.save_dir)
    return trainer
