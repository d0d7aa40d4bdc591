"""Mirror of the reference's repo-root ``val.py``: the tail of ``main_pretrain_encoder.main`` (:35-38) -- after ``worker()``
returned the pre-trained model, fine-tune it once per labelled ratio from the SAME pre-trained weights and score the Dice.

``val`` (val.py:24-42): snapshot the state dict on the CPU, and per ratio: restore it, then -- inside the seeded context --
``_val`` (:45-66): deep copies of the config sections, ``labeled_scan_num`` set in both the data section and the global
config, ``get_data(..., pretrain=False)``, save directory ``<main>/tra/num_labeled_scan_<n>``, ``FineTuneTrainer(model=,
labeled_loader=, unlabeled_loader=, val_loader=, test_loader=, criterion=KL_div(verbose=False), config=global_config,
**trainer_params)`` -- no learning rate of its own: the optimizer and schedule come from ``config["Optim"]`` /
``config["Scheduler"]`` --, ``init()``, ``start_training()``, ``success()``."""
import os
import random
from collections import OrderedDict
from contextlib import contextmanager
from copy import deepcopy as dcopy
from typing import Any, Dict, List

import numpy as np
import torch
from torch import nn

from .contrastyou import success
from .contrastyou.losses.kl import KL_div
from .semi_seg.data.creator import get_data
from .semi_seg.trainers.finetune import FineTuneTrainer


@contextmanager
def switch_model_device(model: nn.Module, device: str = "cpu"):
    previous_device = next(model.parameters()).device
    model.to(device)
    yield
    model.to(previous_device)


@contextmanager
def fix_all_seed_within_context(seed):
    """contrastyou/utils/utils.py:156-173"""
    state = (random.getstate(), np.random.get_state(), torch.random.get_rng_state())
    cuda = torch.cuda.is_available()
    if cuda:
        cuda_state = torch.cuda.get_rng_state_all()
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if cuda:
        torch.cuda.manual_seed_all(seed)
    yield
    random.setstate(state[0])
    np.random.set_state(state[1])
    torch.random.set_rng_state(state[2])
    if cuda:
        torch.cuda.set_rng_state_all(cuda_state)


def val(*, model: nn.Module, save_dir: str, base_config: Dict[str, Any], labeled_ratios: List[float], seed: int = 10):
    with switch_model_device(model, device="cpu"):
        holding_state_dict = OrderedDict((k, v.clone()) for k, v in model.state_dict().items())
    data_params = base_config["Data"]
    loader_l_params = base_config["LabeledLoader"]
    loader_u_params = base_config["UnlabeledLoader"]
    trainer_params = base_config["Trainer"]
    trainers = []
    for ratio in labeled_ratios:
        model.load_state_dict(holding_state_dict)
        with fix_all_seed_within_context(seed):
            trainers.append(_val(model=model, data_params=data_params, labeled_loader_params=loader_l_params,
                                 unlabeled_loader_params=loader_u_params, main_save_dir=save_dir,
                                 trainer_params=trainer_params, global_config=base_config, labeled_data_ratio=ratio))
    return trainers  # (the reference returns None; the trainers carry ``history`` / ``_best_score`` for the caller)


def _val(*, model: nn.Module, labeled_data_ratio: float, data_params: Dict[str, Any],
         labeled_loader_params: Dict[str, Any], unlabeled_loader_params: Dict[str, Any], main_save_dir: str,
         trainer_params: Dict[str, Any], global_config: Dict[str, Any]):
    data_params, trainer_params, global_config = list(map(dcopy, [data_params, trainer_params, global_config]))
    data_params["labeled_scan_num"] = float(labeled_data_ratio)
    global_config["Data"]["labeled_scan_num"] = float(labeled_data_ratio)
    labeled_loader, unlabeled_loader, val_loader, test_loader = get_data(
        data_params=data_params, labeled_loader_params=labeled_loader_params,
        unlabeled_loader_params=unlabeled_loader_params, pretrain=False)
    trainer_params["save_dir"] = os.path.join(main_save_dir, "tra",
                                              f"num_labeled_scan_{len(labeled_loader.dataset.get_scan_list())}")
    trainer = FineTuneTrainer(model=model, labeled_loader=labeled_loader, unlabeled_loader=unlabeled_loader,
                              val_loader=val_loader, test_loader=test_loader, criterion=KL_div(verbose=False),
                              config=global_config, **trainer_params)
    trainer.init()
    trainer.start_training()
    success(save_dir=trainer.save_dir)
    return trainer
