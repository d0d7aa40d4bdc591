"""Synthetic on-device pre-train batches of the shape BASELINE.json names: ``torch.rand`` slices in
[0,1) like ToTensor output, ACDC-like meta-labels (partition = i % 3, scan = i // 3), in the reference's loader
tuple format ((image, image_tf, target, target_tf), filenames, (partitions, groups))."""
import torch


def acdc_like_meta(bs: int, shift: int = 0):
    """``shift`` rotates the batch composition (a different slice order per batch, as a sampler would give)"""
    idx = [(i + shift) % bs for i in range(bs)]
    partitions = [str(i % 3) for i in idx]
    groups = [f"patient{i // 3:03d}_00" for i in idx]
    filenames = [f"patient{i // 3:03d}_00_{i % 3:02d}" for i in idx]
    return filenames, partitions, groups


def prostate_like_meta(bs: int, partition_num: int = 8, shift: int = 0):
    """Prostate-like batch composition (BASELINE.json configs[3]): scans "CaseNN" cut into ``partition_num`` slice
    partitions; group strings "CaseNN_k" (patient id before the underscore, semi_seg/hooks/utils.py:53-56)."""
    idx = [(i + shift) % bs for i in range(bs)]
    partitions = [str(i % partition_num) for i in idx]
    groups = [f"Case{i // partition_num:02d}_{i % partition_num}" for i in idx]
    filenames = [f"Case{i // partition_num:02d}_{i % partition_num:03d}" for i in idx]
    return filenames, partitions, groups


class SyntheticPretrainLoader:
    """Infinite iterator.  ``resident=True`` cycles through ``pool`` DISTINCT pre-generated device batches (inputs already
    in HBM when the timed region starts; every batch has its own images and its own slice order, hence its own label
    vector); otherwise it draws a fresh batch on device every step (two ``torch.rand`` launches per step)."""

    def __init__(self, bs=32, size=224, channels=1, device="cuda", seed=1234, resident=True, meta="acdc", pool=1):
        self.bs, self.size, self.channels, self.device, self.resident = bs, size, channels, device, resident
        self.gen = torch.Generator(device=device).manual_seed(seed)
        self._meta_fn = prostate_like_meta if meta == "prostate" else acdc_like_meta
        self.meta = self._meta_fn(bs)
        self._drawn = 0
        self._pool = [self._draw() for _ in range(max(1, pool))] if resident else None
        self._next = 0

    def _draw(self):
        shape = (self.bs, self.channels, self.size, self.size)
        img = torch.rand(shape, device=self.device, generator=self.gen)
        img_tf = torch.rand(shape, device=self.device, generator=self.gen)
        tgt = torch.zeros((self.bs, 1, 1, 1), dtype=torch.long, device=self.device)
        filenames, partitions, groups = self.meta if self._drawn == 0 else self._meta_fn(self.bs, shift=5 * self._drawn)
        self._drawn += 1
        return (img, img_tf, tgt, tgt), filenames, (partitions, groups)

    def __iter__(self):
        return self

    def __next__(self):
        if not self.resident:
            return self._draw()
        batch = self._pool[self._next]
        self._next = (self._next + 1) % len(self._pool)
        return batch


class SyntheticLabeledLoader:
    """Labelled batches for the fine-tune / evaluation epochers: images in [0,1), integer label maps [B,1,H,W] drawn as
    a few random blobs per class; ``twice=True`` yields the labelled-loader format ((image, image_tf, target,
    target_tf), filenames, (partitions, groups)), else the single-transform format ((image, target), ...) of the
    validation loaders.  ``length`` makes it a finite, re-iterable loader (``len()`` is what EvalEpocher reads)."""

    def __init__(self, bs=8, size=224, channels=1, num_classes=4, device="cuda", seed=99, twice=True, length=None, pool=1):
        self.bs, self.size, self.channels, self.K, self.device = bs, size, channels, num_classes, device
        self.twice, self.length = twice, length
        self.gen = torch.Generator(device=device).manual_seed(seed)
        self.meta = acdc_like_meta(bs)
        # ``pool`` distinct resident batches (own images and label maps), cycled by ``next()``: a benchmark step then sees
        # fresh data every step, as the pre-train loader gives it
        self._pool = [self._draw() for _ in range(max(1, pool))]
        self._batch, self._next = self._pool[0], 0

    def _draw(self):
        img = torch.rand((self.bs, self.channels, self.size, self.size), device=self.device, generator=self.gen)
        coarse = torch.randint(0, self.K, (self.bs, 1, max(1, self.size // 16), max(1, self.size // 16)),
                               device=self.device, generator=self.gen)
        tgt = torch.nn.functional.interpolate(coarse.float(), size=(self.size, self.size), mode="nearest").long()
        filenames, partitions, groups = self.meta
        if self.twice:
            return (img, img, tgt, tgt), filenames, (partitions, groups)
        return (img, tgt), filenames, (partitions, groups)

    def __len__(self):
        if self.length is None:
            raise TypeError("infinite loader")
        return self.length

    def __iter__(self):
        if self.length is None:
            return self
        return iter([self._batch] * self.length)

    def __next__(self):
        batch = self._pool[self._next]
        self._next = (self._next + 1) % len(self._pool)
        return batch
