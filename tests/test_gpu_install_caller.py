"""The drop-in seam, exercised the way the reference's entry point uses it: after
``spcl_amd.install()`` the IMPORT LINES of ``main_pretrain_encoder.py:13-16`` / ``hook_creator.py:1`` resolve to the
HIP-backed mirror, and the BODY of ``worker()`` (``main_pretrain_encoder.py:41-74``, restated below statement by statement
with a stub ``get_data`` -- the data sets are Google-Drive downloads -- and a short schedule) runs: constructor call with
the reference's keyword set, hook creation from the config sections, ``register_hooks``, ``feature_until_from_hooks``,
``model.set_grad(False, start=until, include_start=False)``, ``init()``, ``start_training()``, ``success()``."""
import os
from copy import deepcopy as dcopy

import pytest
import torch

pytestmark = pytest.mark.gpu

CONFIG = {  # config/base.yaml + config/pretrain.yaml + config/hooks/spinfonce.yaml, merged as ConfigManger would
    "RandomSeed": 10,
    "Arch": {"input_dim": 1, "num_classes": 4, "checkpoint": None, "max_channel": 128, "momentum": 0.1},
    "Optim": {"name": "RAdam", "lr": 0.0000001, "weight_decay": 0.00001},
    "Scheduler": {"multiplier": 400, "warmup_max": 10},
    "Data": {"name": "acdc", "labeled_scan_num": 1},
    "LabeledLoader": {"shuffle": True, "batch_size": 5, "num_workers": 5},
    "UnlabeledLoader": {"shuffle": True, "batch_size": 5, "num_workers": 5},
    "Trainer": {"save_dir": "tmp", "device": "cuda", "num_batches": 6, "max_epoch": 3, "two_stage": False,
                "disable_bn": False, "name": None},
    "ContrastiveLoaderParams": {"scan_sample_num": 4, "partition_sample_num": 1, "num_workers": 8},
    "SPInfonceParams": {"feature_names": "Conv5", "weights": 1, "contrast_ons": "partition", "begin_values": 10000,
                        "end_values": 10000, "mode": "soft", "p": 0.5, "correct_grad": True},
}


def test_worker_body_runs_on_the_installed_mirror(tmp_path):
    import spcl_amd
    names = spcl_amd.install()
    for must in ("semi_seg.arch", "semi_seg.hooks", "semi_seg.hooks.creator", "semi_seg.trainers.new_pretrain",
                 "semi_seg.epochers.new_pretrain", "contrastyou.losses.contrast_loss3", "hook_creator"):
        assert must in names, must
    # ---- the reference's import lines (main_pretrain_encoder.py:5,11-16; hook_creator.py:1)
    from deepclustering2.loss import KL_div
    from contrastyou import success
    from hook_creator import create_hook_from_config
    from semi_seg.arch import UNet
    from semi_seg.hooks import feature_until_from_hooks
    from semi_seg.hooks import create_infonce_hooks, create_sp_infonce_hooks, create_discrete_mi_consistency_hook  # noqa
    from semi_seg.trainers.new_pretrain import PretrainEncoderTrainer
    from semi_seg.epochers.new_pretrain import PretrainEncoderEpocher  # noqa
    from contrastyou.losses.contrast_loss3 import SelfPacedSupConLoss
    from semi_seg.data import synthetic_slice_store
    assert UNet.__module__.startswith("spcl_amd.") and PretrainEncoderTrainer.__module__.startswith("spcl_amd.")

    def get_data(data_params, labeled_loader_params, unlabeled_loader_params, pretrain=False, total_freedom=False):
        """stub of semi_seg/data/creator.py:155-161: four loaders over a synthetic ACDC-shaped store"""
        store = synthetic_slice_store(scans=6, slices_per_scan=(6, 9), size=64, device="cuda", seed=1)

        class Loader:
            dataset = store
        return Loader(), Loader(), Loader(), Loader()

    absolute_save_dir, seed = str(tmp_path), 10
    # ---- worker(), main_pretrain_encoder.py:41-74 (only the save_dir join and the seeding context are simplified)
    config = dcopy(CONFIG)
    model_checkpoint = config["Arch"].pop("checkpoint", None)
    torch.manual_seed(seed)
    model = UNet(**config["Arch"])
    assert not model_checkpoint
    labeled_loader, unlabeled_loader, val_loader, test_loader = get_data(
        data_params=config["Data"], labeled_loader_params=config["LabeledLoader"],
        unlabeled_loader_params=config["UnlabeledLoader"], pretrain=True, total_freedom=True)
    trainer = PretrainEncoderTrainer(model=model, labeled_loader=labeled_loader, unlabeled_loader=unlabeled_loader,
                                     val_loader=val_loader, test_loader=test_loader,
                                     criterion=KL_div(verbose=False), config=config,
                                     save_dir=os.path.join(absolute_save_dir, "pre"),
                                     **{k: v for k, v in config["Trainer"].items() if k != "save_dir"})
    trainer._contrastive_loader._views.out_hw = (32, 32)  # (test only: small crops of the 64x64 synthetic slices)
    hooks = create_hook_from_config(model, config, is_pretrain=True)
    assert len(hooks) > 0, "void hooks"
    trainer.register_hooks(*hooks)
    until = feature_until_from_hooks(*hooks)
    assert until == "Conv5"
    trainer.forward_until = until
    w0 = model._Conv5.conv[0].weight.detach().clone()
    d0 = model._Up5.up[1].weight.detach().clone()
    with model.set_grad(False, start=until, include_start=False):
        trainer.init()
        trainer.start_training()
    success(save_dir=trainer.save_dir)
    # the loop the seam hands out replays its step from a hipGraph (stepgraph.py): 2 eager steps, capture, 4 replays
    sg = trainer._last_epocher._step_graph
    assert sg is not None and sg.captured and not sg.failed and sg.replays == 4
    # ---- what the run must have left behind
    assert os.path.exists(os.path.join(trainer.save_dir, ".success"))
    assert os.path.exists(os.path.join(trainer.save_dir, "last.pth"))
    assert os.path.exists(os.path.join(trainer.save_dir, "config.yaml"))
    assert len(trainer.history) == 2  # epochs 1 and 2 of range(max(0 + 1, 0), 3) (new_pretrain.py:70-72)
    stats = trainer.history[-1]
    hook_group = [g for g in stats if g != "semi"]
    assert len(hook_group) == 1 and "spinfonce" in hook_group[0]  # the hook's meters live under its name (hooks/utils.py:68-74)
    loss = stats[hook_group[0]]["loss"]["mean"]
    assert loss == loss and loss > 0 and stats["semi"]["reg_loss"]["mean"] > 0
    assert stats[hook_group[0]]["age_param"]["mean"] == 10000 and 0 < stats[hook_group[0]]["sp_weight"]["mean"] <= 1
    assert not torch.equal(w0.cuda(), model._Conv5.conv[0].weight.detach())  # the encoder moved
    assert torch.equal(d0.cuda(), model._Up5.up[1].weight.detach())          # the frozen decoder did not
    assert all(p.requires_grad for p in model._Up5.parameters())              # set_grad restored on exit
    crit = hooks[0]._hooks[0]._criterion
    assert isinstance(crit, SelfPacedSupConLoss) and crit.age_param == 10000
    with pytest.raises(RuntimeError):  # trainer/base.py:54-55
        trainer.register_hooks(*hooks)
    # ---- resume: continues AFTER the saved epoch (trainer/base.py:95)
    t2 = PretrainEncoderTrainer(model=UNet(**config["Arch"]), labeled_loader=labeled_loader,
                                unlabeled_loader=unlabeled_loader, val_loader=val_loader, test_loader=test_loader,
                                criterion=KL_div(verbose=False), config=config, save_dir=os.path.join(absolute_save_dir, "re"),
                                **{**{k: v for k, v in config["Trainer"].items() if k != "save_dir"}, "max_epoch": 4})
    t2._contrastive_loader._views.out_hw = (32, 32)
    h2 = create_hook_from_config(t2._model, config, is_pretrain=True)
    t2.register_hooks(*h2)
    t2.forward_until = "Conv5"
    with t2._model.set_grad(False, start="Conv5", include_start=False):
        t2.init()
        t2.resume_from_path(os.path.join(trainer.save_dir, "last.pth"))
        assert t2._cur_epoch == 2
        t2.start_training()
    assert len(t2.history) == 1 and t2._cur_epoch == 3  # only epoch 3 was left

    cfg_missing = dcopy(CONFIG)
    cfg_missing.pop("ContrastiveLoaderParams")
    with pytest.raises(RuntimeError):  # new_pretrain.py:38-40
        PretrainEncoderTrainer(model=model, labeled_loader=labeled_loader, unlabeled_loader=unlabeled_loader,
                               val_loader=val_loader, test_loader=test_loader, criterion=None, config=cfg_missing,
                               save_dir=str(tmp_path / "x"))
