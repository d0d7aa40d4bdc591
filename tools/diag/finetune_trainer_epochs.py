"""Wall clock of whole EPOCHS of the real FineTuneTrainer as ``val()`` drives it (val.py:24-66 with config/base.yaml: 200 batches
of 5 labelled slices per epoch, then the validation and the test pass, one scan per batch) on a synthetic ACDC-shaped labelled
store at 256^2 -> 224^2 crops; `sync` as second argument prints where the host waits for the device."""
import os
import sys
import tempfile
import time
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spcl_amd  # noqa: E402

spcl_amd.install()
from semi_seg.data import ACDCSliceStore  # noqa: E402
from semi_seg.data.creator import register_dataset  # noqa: E402
from semi_seg.arch import UNet  # noqa: E402
from val import val  # noqa: E402

EPOCHS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
CONFIG = {
    "RandomSeed": 10,
    "Arch": {"input_dim": 1, "num_classes": 4, "checkpoint": None, "max_channel": 256, "momentum": 0.1},
    "Optim": {"name": "RAdam", "lr": 0.0000001, "weight_decay": 0.00001},
    "Scheduler": {"multiplier": 300, "warmup_max": 10},
    "Data": {"name": "acdc", "labeled_scan_num": 1},
    "LabeledLoader": {"shuffle": True, "batch_size": 5, "num_workers": 5},
    "UnlabeledLoader": {"shuffle": True, "batch_size": 5, "num_workers": 5},
    "Trainer": {"save_dir": "tmp", "device": "cuda", "num_batches": 200, "max_epoch": EPOCHS, "two_stage": False,
                "disable_bn": False, "name": None},
}


def store(scans, seed, size=256):
    g = torch.Generator().manual_seed(seed)
    imgs, names = [], []
    for s, scan in enumerate(scans):
        base = torch.nn.functional.interpolate(torch.rand(1, 1, 8, 8, generator=g), size=(size, size), mode="bilinear",
                                               align_corners=False)[0, 0]
        for k in range(6 + (s % 7)):
            imgs.append(torch.round((base * (0.7 + 0.02 * k)).clamp(0, 1) * 255) / 255)
            names.append(f"{scan}_{k:02d}")
    images = torch.stack(imgs)
    targets = (images * 255 / 52).floor().clamp(0, 3).to(torch.uint8)
    return ACDCSliceStore(images.cuda(), names, targets=targets.cuda())


TRAIN = [f"patient{p:03d}_{e:02d}" for p in list(range(1, 50)) + [100] for e in (0, 1)]   # (holds the predefined labelled scans)
TEST = [f"patient{p:03d}_{e:02d}" for p in range(101, 151) for e in (0, 1)]               # 100 scans: 35 validate, 65 test
register_dataset("acdc", lambda mode: store(TRAIN, 3) if mode == "train" else store(TEST, 4))
torch.manual_seed(10)
model = UNet(**{k: v for k, v in CONFIG["Arch"].items() if k != "checkpoint"}).cuda()
model.set_compute_dtype(torch.bfloat16)
if len(sys.argv) > 2 and sys.argv[2] == "sync":
    torch.cuda.set_sync_debug_mode("warn")
    warnings.simplefilter("default")
from semi_seg.trainers.new_trainer import FineTuneTrainer as _FT  # noqa: E402

stamps, T = [], _FT
orig = T.save_to


def timed(self, name, *a, **k):
    out = orig(self, name, *a, **k)
    if name == "last.pth":
        torch.cuda.synchronize()
        stamps.append(time.perf_counter())
    return out


T.save_to = timed
t0 = time.perf_counter()
trainers = val(model=model, save_dir=tempfile.mkdtemp(prefix="spcl_ft_"), base_config=CONFIG, seed=10, labeled_ratios=[2])
stamps = [t0] + stamps
for i in range(1, len(stamps)):
    print(f"epoch {i}: {1e3 * (stamps[i] - stamps[i - 1]):.1f} ms", flush=True)
print("val scans", len(trainers[0]._val_loader), "test scans", len(trainers[0]._test_loader), "score", trainers[0].history[-1]["score"])
