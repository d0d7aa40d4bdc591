"""Loaders of the contrastive pre-train loop (``semi_seg/trainers/_helper.py:31-74``): ACDC batches are composed by
``ContrastBatchSampler``, the other data sets by an infinite random permutation with batch size ``scan_sample_num x
partition_num``; every batch is gathered + augmented on the device and handed over in the reference's tuple format
``((image, image_tf, target, target_tf), filenames, (partitions, scans))``."""
import random
from typing import Iterator

import torch

from .augment import RECIPES, PretrainViews, RecipeViews, redraw_jitter
from .rearr import ContrastBatchSampler


class InfiniteRandomSampler:
    """``contrastyou/data/sampler.py:203-228``: endless stream of random permutations of the data set"""

    def __init__(self, data_source, shuffle=True):
        self._n, self._shuffle = len(data_source), shuffle

    def __iter__(self) -> Iterator[int]:
        while True:
            order = list(range(self._n))
            if self._shuffle:
                random.shuffle(order)
            yield from order


class ContrastiveDeviceLoader:
    """infinite iterator of device batches"""

    def __init__(self, store, *, batch_sampler=None, sampler=None, batch_size=None, out_hw=(224, 224), recipe=None,
                 **legacy_recipe):
        """``recipe``: a key of ``augment.RECIPES`` (default: the store's own pre-train recipe -- ACDCStrongTransforms.pretrain /
        ProstateStrongTransforms.pretrain, semi_seg/augment.py:6-22,54-69, images rotated with BILINEAR as the reference's
        wrapper selects) or a dict of the same keys; ``legacy_recipe`` keywords (``pil_exact=``, ``degrees=`` ...) select the
        round-4 ``PretrainViews`` (nearest image rotation) instead."""
        assert (batch_sampler is None) != (sampler is None)
        self.dataset = store
        if legacy_recipe:
            self._views = PretrainViews(store.images, out_hw, **legacy_recipe)
        else:
            name = recipe or ("prostate_pretrain" if getattr(store, "data_name", "acdc") == "prostate" else "acdc_pretrain")
            self._views = RecipeViews(store.images, name, out_hw)
        self._batch_sampler, self._sampler, self._batch_size = batch_sampler, sampler, batch_size
        self._it = None

    def _index_batches(self):
        if self._batch_sampler is not None:
            yield from iter(self._batch_sampler)
        else:
            it = iter(self._sampler)
            while True:
                yield [next(it) for _ in range(self._batch_size)]

    def __iter__(self):
        self._it = self._index_batches()
        return self

    def __next__(self):
        if self._it is None:
            self._it = self._index_batches()
        idx = next(self._it)
        img, img_tf = self._views(idx)
        metas = [self.dataset.meta(i) for i in idx]
        tgt = torch.zeros(len(idx), 1, 1, 1, dtype=torch.long, device=img.device)  # pre-training never reads the labels
        return (img, img_tf, tgt, tgt), [m[0] for m in metas], ([m[1] for m in metas], [m[2] for m in metas])


class LabeledDeviceLoader:
    """the labelled loader of the fine-tune loop (``FineTuneEpocher._run_only_label``, semi_seg/epochers/new_epocher.py:260-283)
    on device: batches ``((image, image_tf, target, target_tf), filenames, (partitions, scans))`` from a store WITH label maps
    through a recipe of ``augment.RECIPES`` -- by default the data set's `label` recipe (``ACDCStrongTransforms.label``,
    semi_seg/augment.py:23-34: RandomCrop, then RandomRotation(30)); ``semi_seg.data.creator`` passes the `pretrain` recipe,
    which is what the reference's ``get_data`` trains on (creator.py:30).  Image (BILINEAR) and label map (NEAREST) of a pair
    share one geometry.  The second pair follows ``SequentialWrapperTwice`` (contrastyou/augment/synchronize.py:129-150):
    ``total_freedom=True`` (its default, the `label` recipes keep it) -- an independent draw of the whole recipe;
    ``False`` -- the first pair's geometry with a fresh draw of the image-only colour jitter (the same tensors when the
    recipe has none)."""

    def __init__(self, store, *, batch_size, sampler=None, out_hw=(224, 224), recipe=None, total_freedom=True):
        if getattr(store, "targets", None) is None:
            raise ValueError("LabeledDeviceLoader: the store has no label maps (DeviceSliceStore(..., targets=))")
        self.dataset, self._batch_size = store, int(batch_size)
        name = recipe or ("prostate_label" if getattr(store, "data_name", "acdc") == "prostate" else "acdc_label")
        rec = dict(RECIPES[name]) if isinstance(name, str) else dict(name)
        if rec.get("resize"):
            raise NotImplementedError("resize the store (images BILINEAR, label maps NEAREST) before building the loader")
        self._views = RecipeViews(store.images, rec, out_hw, labels=store.targets)
        self._sampler = sampler if sampler is not None else InfiniteRandomSampler(store, shuffle=True)
        self._total_freedom = bool(total_freedom)
        self._it = None

    def __iter__(self):
        self._it = iter(self._sampler)
        return self

    def __next__(self):
        if self._it is None:
            self._it = iter(self._sampler)
        idx = [next(self._it) for _ in range(self._batch_size)]
        rows = self._views.rows(idx)
        if self._total_freedom:
            rows2 = self._views.rows(idx)
        elif self._views.recipe.get("brightness") or self._views.recipe.get("contrast"):
            rows2 = [redraw_jitter(r, self._views.recipe) for r in rows]
        else:
            rows2 = None
        n = len(idx)
        img, tgt = self._views.apply(rows + (rows2 or []), with_labels=True)  # (both pairs in one launch)
        img2, tgt2 = (img[n:], tgt[n:]) if rows2 is not None else (img, tgt)
        metas = [self.dataset.meta(i) for i in idx]
        return (img[:n], img2, tgt[:n], tgt2), [m[0] for m in metas], ([m[1] for m in metas], [m[2] for m in metas])


def get_contrastive_dataloader(partial_loader, contrastive_params, device="cuda", out_hw=(224, 224)):
    """``_get_contrastive_dataloader`` (semi_seg/trainers/_helper.py:31-74): a loader over ALL training scans of the data
    set behind ``partial_loader`` (its ``.dataset``, or the store itself) -> (contrastive loader, monitor loader).
    ``num_workers`` is accepted and ignored: there are no worker processes on this path."""
    params = dict(contrastive_params)
    params.pop("num_workers", None)
    store = getattr(partial_loader, "dataset", partial_loader)
    if not hasattr(store, "images") or not hasattr(store, "meta"):
        raise TypeError("get_contrastive_dataloader: expected a semi_seg.data.DeviceSliceStore (or a loader whose "
                        f"`.dataset` is one), got {type(store).__name__}")
    if getattr(store, "data_name", "acdc") == "acdc":  # "only group the acdc dataset" (_helper.py:58-63)
        loader = ContrastiveDeviceLoader(store, batch_sampler=ContrastBatchSampler(store, **params), out_hw=out_hw)
    else:
        loader = ContrastiveDeviceLoader(store, sampler=InfiniteRandomSampler(store, shuffle=True),
                                         batch_size=params["scan_sample_num"] * store.partition_num, out_hw=out_hw)
    return loader, None
