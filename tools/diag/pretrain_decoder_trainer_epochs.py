"""(decoder variant, see pretrain_trainer_epochs.py) Wall clock of whole EPOCHS of the real PretrainDecoderTrainer (the body of main_pretrain_encoder.worker with the reference's
config: 200 batches per epoch of 10 scans x 3 partitions, self-paced InfoNCE on Conv5, checkpoint every epoch) on a synthetic
ACDC-shaped device store: what a user's training run pays per epoch, checkpoint writes and all."""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spcl_amd  # noqa: E402

spcl_amd.install()
from deepclustering2.loss import KL_div  # noqa: E402
from hook_creator import create_hook_from_config  # noqa: E402
from semi_seg.arch import UNet  # noqa: E402
from semi_seg.data import synthetic_slice_store  # noqa: E402
from semi_seg.hooks import feature_until_from_hooks  # noqa: E402
from semi_seg.trainers.new_pretrain import PretrainDecoderTrainer as PretrainEncoderTrainer  # noqa: E402

CONFIG = {
    "RandomSeed": 10,
    "Arch": {"input_dim": 1, "num_classes": 4, "checkpoint": None, "max_channel": 256, "momentum": 0.1},
    "Optim": {"name": "RAdam", "lr": 0.0000001, "weight_decay": 0.00001},
    "Scheduler": {"multiplier": 400, "warmup_max": 10},
    "Data": {"name": "acdc", "labeled_scan_num": 1},
    "LabeledLoader": {"shuffle": True, "batch_size": 5, "num_workers": 5},
    "UnlabeledLoader": {"shuffle": True, "batch_size": 5, "num_workers": 5},
    "Trainer": {"save_dir": "tmp", "device": "cuda", "num_batches": 200, "max_epoch": int(sys.argv[1]) if len(sys.argv) > 1 else 5,
                "two_stage": False, "disable_bn": False, "name": None},
    "ContrastiveLoaderParams": {"scan_sample_num": 10, "partition_sample_num": 1, "num_workers": 8},
    "InfonceParams": {"feature_names": ["Conv5", "Up_conv3", "Up_conv2"], "weights": [1, 0.5, 0.25],
                      "contrast_ons": ["partition", "partition", "partition"]},
}
store = synthetic_slice_store(scans=100, slices_per_scan=(9, 12), size=256, device="cuda", seed=1)


class Loader:
    dataset = store


torch.manual_seed(10)
model = UNet(**{k: v for k, v in CONFIG["Arch"].items() if k != "checkpoint"})
model.set_compute_dtype(torch.bfloat16)
save = tempfile.mkdtemp(prefix="spcl_pre_")
trainer = PretrainEncoderTrainer(model=model, labeled_loader=Loader(), unlabeled_loader=Loader(), val_loader=Loader(),
                                 test_loader=Loader(), criterion=KL_div(verbose=False), config=CONFIG,
                                 save_dir=os.path.join(save, "pre"),
                                 **{k: v for k, v in CONFIG["Trainer"].items() if k != "save_dir"})
hooks = create_hook_from_config(model, CONFIG, is_pretrain=True)
trainer.register_hooks(*hooks)
trainer.forward_until = feature_until_from_hooks(*hooks)
stamps = [time.perf_counter()]
orig_save = trainer.save_to
save_s, mem = [], []


def timed_save(*a, **k):
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = orig_save(*a, **k)
    save_s.append(time.perf_counter() - t)
    stamps.append(time.perf_counter())
    mem.append((torch.cuda.memory_reserved() / 2 ** 30, torch.cuda.memory_allocated() / 2 ** 30))
    return out


trainer.save_to = timed_save
import contextlib
with contextlib.ExitStack() as es:
    es.enter_context(model.set_grad(False))
    es.enter_context(model.set_grad(True, start="Conv5", end=trainer.forward_until, include_start=False))
    trainer.init()
    stamps[0] = time.perf_counter()
    trainer.start_training()
torch.cuda.synchronize()
for i in range(1, len(stamps)):
    print(f"epoch {i}: {1e3 * (stamps[i] - stamps[i - 1]):.1f} ms of which checkpoint write {1e3 * save_s[i - 1]:.1f} ms "
          f"({6000 / (stamps[i] - stamps[i - 1]) / 1e3:.1f} k slices/s); device memory reserved {mem[i - 1][0]:.2f} GiB, "
          f"allocated {mem[i - 1][1]:.2f} GiB", flush=True)
