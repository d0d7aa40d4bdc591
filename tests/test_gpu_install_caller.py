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


def _worker(tmp_path, config_in):
    """``worker()`` (main_pretrain_encoder.py:41-74) on the installed mirror, statement by statement"""
    import spcl_amd
    names = spcl_amd.install()
    for must in ("semi_seg.arch", "semi_seg.hooks", "semi_seg.hooks.creator", "semi_seg.trainers.new_pretrain",
                 "semi_seg.epochers.new_pretrain", "contrastyou.losses.contrast_loss3", "hook_creator", "val",
                 "semi_seg.data.creator", "semi_seg.trainers.new_trainer"):
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
    config = dcopy(config_in)
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
    return model, trainer, hooks, config, (labeled_loader, unlabeled_loader, val_loader, test_loader), (w0, d0)


def test_worker_body_runs_on_the_installed_mirror(tmp_path):
    model, trainer, hooks, config, loaders, (w0, d0) = _worker(tmp_path, CONFIG)
    labeled_loader, unlabeled_loader, val_loader, test_loader = loaders
    absolute_save_dir = str(tmp_path)
    from deepclustering2.loss import KL_div
    from hook_creator import create_hook_from_config
    from semi_seg.arch import UNet
    from semi_seg.trainers.new_pretrain import PretrainEncoderTrainer
    from contrastyou.losses.contrast_loss3 import SelfPacedSupConLoss
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


# ---- the tail of main() (main_pretrain_encoder.py:35-38): val(model=worker(...), base_config=..., labeled_ratios=ratio_zoo[..])
BASE_CONFIG = {  # config/base.yaml alone (``separate_pretrain_finetune_configs``: the fine-tune stage never sees pretrain.yaml)
    **{k: dcopy(v) for k, v in CONFIG.items() if k not in ("ContrastiveLoaderParams", "SPInfonceParams")},
    "Scheduler": {"multiplier": 300, "warmup_max": 10},
    "Trainer": {"save_dir": "tmp", "device": "cuda", "num_batches": 5, "max_epoch": 2, "two_stage": False,
                "disable_bn": False, "name": None},
}
TRAIN_SCANS = ["patient100_00", "patient027_01", "patient038_01", "patient067_01", "patient003_00", "patient011_01",
               "patient050_00", "patient051_01"]
VAL_SCANS = ["patient150_00", "patient151_01", "patient152_00", "patient153_01", "patient154_00", "patient155_01"]


def _labelled_store(scans, seed, size=64):
    from semi_seg.data import ACDCSliceStore
    g = torch.Generator().manual_seed(seed)
    imgs, names = [], []
    for s in scans:
        base = torch.nn.functional.interpolate(torch.rand(1, 1, 6, 6, generator=g), size=(size, size), mode="bilinear",
                                               align_corners=False)[0, 0]
        for k in range(6):
            imgs.append(torch.round((base * (0.7 + 0.05 * k)).clamp(0, 1) * 255) / 255)
            names.append(f"{s}_{k:02d}")
    images = torch.stack(imgs)
    targets = (images * 255 / 52).floor().clamp(0, 3).to(torch.uint8)  # four classes from the grey level
    return ACDCSliceStore(images.cuda(), names, targets=targets.cuda())


def test_main_tail_val_finetunes_from_the_config(tmp_path):
    """``val()`` / ``_val()`` (val.py:24-66) on the installed mirror with the model ``worker()`` returned: per labelled
    ratio the pre-trained weights are restored, the reference's predefined labelled scans are selected, and the fine-tune
    trainer takes its optimizer and schedule FROM THE CONFIG (contrastyou/trainer/base.py:60-83):
    lr = config lr x the warm-up factor towards ``multiplier`` = 300."""
    model, pre_trainer, _, _, _, _ = _worker(tmp_path, CONFIG)
    from val import val
    from semi_seg import ratio_zoo
    from semi_seg.data.creator import register_dataset
    from semi_seg.trainers.new_trainer import FineTuneTrainer
    assert ratio_zoo["acdc"] == [1, 2, 4, 174]
    register_dataset("acdc", lambda mode: _labelled_store(TRAIN_SCANS if mode == "train" else VAL_SCANS,
                                                          seed=3 if mode == "train" else 4), out_hw=(32, 32))
    pre = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    restored = []
    orig_load = model.load_state_dict
    model.load_state_dict = lambda sd, *a, **kw: (restored.append({k: v.clone() for k, v in sd.items()}), orig_load(sd, *a, **kw))[1]
    base_config = dcopy(BASE_CONFIG)
    from semi_seg.data import creator as _creator
    try:
        trainers = val(model=model, save_dir=str(tmp_path), base_config=base_config, seed=10,
                       labeled_ratios=ratio_zoo["acdc"][:2])
    finally:  # (the registry is module state: leave it as it was found)
        _creator._FACTORIES.pop("acdc", None)
        _creator._OUT_HW["acdc"] = (224, 224)
    assert base_config == BASE_CONFIG  # _val works on deep copies (val.py:48)
    assert len(trainers) == 2 and len(restored) == 2
    for sd in restored:  # every ratio starts from the SAME pre-trained weights (val.py:34)
        assert all(torch.equal(sd[k].cpu(), pre[k]) for k in pre)
    for n, tr in zip((1, 2), trainers):
        assert isinstance(tr, FineTuneTrainer)
        assert tr.save_dir == os.path.join(str(tmp_path), "tra", f"num_labeled_scan_{n}")
        assert sorted(tr._labeled_loader.dataset.get_scan_list()) == sorted(TRAIN_SCANS[:1] if n == 1 else ["patient027_01", "patient100_00"])
        assert not set(tr._val_loader.dataset.get_scan_list()) & set(tr._test_loader.dataset.get_scan_list())
        assert len(tr._val_loader) == 2 and len(tr._test_loader) == 4  # int(6 x 0.35) scans validate, the rest test
        for f in (".success", "last.pth", "config.yaml"):
            assert os.path.exists(os.path.join(tr.save_dir, f)), f
        # optimizer and schedule: config/base.yaml's, not keyword defaults
        assert tr._optim_cfg == {"lr": 0.0000001, "weight_decay": 0.00001} and tr._optim_name == "RAdam"
        assert tr._sched_cfg == {"multiplier": 300, "warmup_max": 10}
        assert tr._config["Data"]["labeled_scan_num"] == float(n)
        lrs = [h["tra"]["semi"]["lr"]["mean"] for h in tr.history]
        want = [1e-7 * ((300 - 1.0) * e / 10 + 1.0) for e in (0, 1)]  # epochs 1, 2 run at warm-up steps 0, 1
        assert len(lrs) == 2 and all(abs(a - b) < 1e-12 for a, b in zip(lrs, want)), (lrs, want)
        assert abs(tr._optimizer.param_groups[0]["lr"] - 1e-7 * (299.0 * 2 / 10 + 1.0)) < 1e-12
        assert all(0.0 <= h["score"] <= 1.0 and "DSC_mean" in h["val"]["eval"]["dice"] for h in tr.history)
    assert all(p.requires_grad for p in model.parameters())  # the decoder trains in this stage
    # keyword overrides still win over the config; a config without a Scheduler section builds none (trainer/base.py:72-73)
    t = FineTuneTrainer(model=model, labeled_loader=trainers[0]._labeled_loader, val_loader=trainers[0]._val_loader,
                        criterion=None, config={"Optim": {"name": "RAdam", "lr": 2e-7, "weight_decay": 1e-4, "ft_lr": 1, "pre_lr": 2}},
                        multiplier=None, lr=3e-7)
    assert t._optim_cfg == {"lr": 3e-7, "weight_decay": 1e-4} and t._sched_cfg is None
    t.init()
    assert t._scheduler is None and t._optimizer.param_groups[0]["lr"] == 3e-7
