"""Mirror of ``semi_seg/data/creator.py``: the four loaders a driver asks for -- ``get_data(data_params=,
labeled_loader_params=, unlabeled_loader_params=, pretrain=, total_freedom=)`` (:155-161) -> labelled, unlabelled,
validation, test -- over device-resident slice stores instead of PNG data sets behind DataLoader workers.

What is kept from the reference (it decides WHICH slices train and which score the Dice):
  * ``create_dataset`` (:27-35): train set + `val`-mode set of the named data set, disjoint scans;
  * ``split_dataset`` (:58-84): sorted scan list, ``numpy.random.permutation`` under seed ``seed`` (the surrounding RNG
    states are restored), cut at the running sums of the ratios;
  * ``split_dataset_with_predefined_filenames`` (:38-55): the fixed labelled scans of ``semi_seg.labeled_filenames`` --
    ``ValueError`` for a scan count without a list, ``KeyError`` for a data set without lists;
  * ``get_data_loaders`` (:97-141): ``labeled_scan_num`` > number of training scans -> ``RuntimeError``; ``pretrain``
    forces a 0.5 split; an empty labelled set -> ``RuntimeError``; scan-grouped test batches;
  * ``create_val_loader`` (:144-152): the `val`-mode set split 0.35 / 0.65 into validation and test, both scan-grouped;
  * ``get_data`` runs under seed 1 (``@fix_seed``, contrastyou/utils/utils.py:206-213).

The data sets themselves are Google-Drive downloads (semi_seg/data/dataset.py:17): a store comes from ``set_data_root``'s
folder (``<root>/ACDC_contrast/{train,val}/{img,gt}/*.png``: the reference's layout) or from a factory registered with
``register_dataset`` (tests, synthetic runs)."""
import os
import random
from typing import Callable, Dict, List, Sequence

import numpy as np
import torch

from .augment import RecipeViews
from .dataset import ACDCSliceStore, DeviceSliceStore, ProstateSliceStore
from .loader import InfiniteRandomSampler, LabeledDeviceLoader

__all__ = ["create_dataset", "create_val_loader", "get_data_loaders", "get_data", "register_dataset", "set_data_root",
           "split_dataset", "split_dataset_with_predefined_filenames", "extract_sub_dataset_based_on_scan_names",
           "ScanBatchSampler", "ScanBatchLoader", "UnlabeledDeviceLoader", "labeled_filenames"]

# semi_seg/__init__.py:76-88 (the labelled scans of the published splits)
labeled_filenames = {
    "acdc": {1: ["patient100_00"],
             2: ["patient027_01", "patient100_00"],
             4: ["patient027_01", "patient038_01", "patient067_01", "patient100_00"],
             8: ["patient027_01", "patient038_01", "patient067_01", "patient100_00", "patient002_00", "patient004_00",
                 "patient006_01", "patient007_00"]},
    "prostate": {3: ["Case10", "Case17", "Case45"],
                 5: ["Case00", "Case10", "Case17", "Case37", "Case45"],
                 7: ["Case00", "Case10", "Case17", "Case34", "Case37", "Case38", "Case45"]},
    "mmwhsct": {1: ["1003"],
                2: ["1003", "1010"]},
}

_FOLDERS = {"acdc": ("ACDC_contrast", ACDCSliceStore), "prostate": ("PROSTATE", ProstateSliceStore)}
_FACTORIES: Dict[str, Callable[[str], DeviceSliceStore]] = {}
_DATA_ROOT = [os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))),
                           ".data")]  # contrastyou/__init__.py: DATA_PATH = PROJECT_PATH / ".data"
_OUT_HW = {"acdc": (224, 224), "prostate": (224, 224)}  # the crops of semi_seg/augment.py


def set_data_root(path: str):
    _DATA_ROOT[0] = str(path)


def register_dataset(name: str, factory: Callable[[str], DeviceSliceStore], out_hw=None):
    """``factory(mode)`` -> store of mode "train" / "val" (train stores with label maps); replaces the folder lookup"""
    _FACTORIES[name] = factory
    if out_hw is not None:
        _OUT_HW[name] = tuple(out_hw)


class _seeded:
    """``fix_all_seed_within_context`` (contrastyou/utils/utils.py:156-173) for the host generators a split draws from"""

    def __init__(self, seed):
        self._seed = seed

    def __enter__(self):
        self._state = (random.getstate(), np.random.get_state(), torch.random.get_rng_state())
        random.seed(self._seed)
        np.random.seed(self._seed)
        torch.manual_seed(self._seed)

    def __exit__(self, *exc):
        random.setstate(self._state[0])
        np.random.set_state(self._state[1])
        torch.random.set_rng_state(self._state[2])
        return False


def create_dataset(name: str, total_freedom: bool = True):
    if name in _FACTORIES:
        tra_set, test_set = _FACTORIES[name]("train"), _FACTORIES[name]("val")
    elif name in _FOLDERS:
        folder, cls = _FOLDERS[name]
        root = os.path.join(_DATA_ROOT[0], folder)
        if not os.path.isdir(os.path.join(root, "train", "img")):
            raise FileNotFoundError(f"{name}: no data under {root} (the reference downloads it; here: set_data_root(...) or "
                                    f"register_dataset({name!r}, factory))")
        tra_set, test_set = cls.from_folder(os.path.join(root, "train")), cls.from_folder(os.path.join(root, "val"))
    else:
        raise KeyError(name)
    assert set(tra_set.get_scan_list()) & set(test_set.get_scan_list()) == set()
    tra_set.total_freedom = total_freedom  # creator.py:31 (``tra_transform._total_freedom``)
    return tra_set, test_set


def extract_sub_dataset_based_on_scan_names(dataset: DeviceSliceStore, group_names: Sequence[str]) -> DeviceSliceStore:
    """contrastyou/data/dataset/base.py:204-227: the slices of the named scans, in the store's order"""
    available = sorted(set(dataset.get_scan_list()))
    for g in group_names:
        assert g in available, (g, available)
    wanted = set(group_names)
    keep = [i for i, s in enumerate(dataset.show_scan_names()) if s in wanted]
    idx = torch.tensor(keep, dtype=torch.long, device=dataset.images.device)
    sub = type(dataset)(dataset.images.index_select(0, idx), [dataset._filenames[i] for i in keep],
                        dict(dataset._scan_info),
                        targets=dataset.targets.index_select(0, idx) if dataset.targets is not None else None)
    sub.total_freedom = getattr(dataset, "total_freedom", True)
    assert set(sub.get_scan_list()) == wanted
    return sub


def split_dataset_with_predefined_filenames(dataset, data_name: str, labeled_ratio: float):
    if data_name not in labeled_filenames:
        raise KeyError(data_name)
    filenames = labeled_filenames[data_name]
    labeled_num = int(len(dataset.get_scan_list()) * labeled_ratio)
    if labeled_num not in filenames:
        raise ValueError(f"{labeled_num} is not defined for `load_predefined_list`, "
                         f"given only {','.join([str(x) for x in filenames.keys()])}")
    labeled_scans = filenames[labeled_num]
    unlabeled_scans = sorted(set(dataset.get_scan_list()) - set(labeled_scans))
    return [extract_sub_dataset_based_on_scan_names(dataset, labeled_scans),
            extract_sub_dataset_based_on_scan_names(dataset, unlabeled_scans)]


def split_dataset(dataset, *ratios: float, seed: int = 1) -> List[DeviceSliceStore]:
    assert sum(ratios) <= 1, ratios
    scan_list = sorted(set(dataset.get_scan_list()))
    with _seeded(seed):
        permuted = np.random.permutation(scan_list).tolist()
    cuts, acc = [], 0.0
    for r in ratios:
        acc += r
        cuts.append(int(len(scan_list) * acc))
    edges = [0] + cuts + [len(scan_list)]
    subs = [extract_sub_dataset_based_on_scan_names(dataset, permuted[a:b]) for a, b in zip(edges[:-1], edges[1:])]
    assert sum(len(set(x.get_scan_list())) for x in subs) == len(scan_list)
    return subs


class ScanBatchSampler:
    """``contrastyou/data/sampler.py:249-284``: one batch = all slices of one scan, scans in first-seen order"""

    def __init__(self, dataset, shuffle=False, is_infinite: bool = False):
        names = dataset.show_scan_names()
        assert len(set(names)) < len(names)
        self.idx_map: Dict[str, List[int]] = {}
        for i, s in enumerate(names):
            self.idx_map.setdefault(s, []).append(i)
        self._shuffle, self._infinite = shuffle, is_infinite

    def __len__(self):
        return len(self.idx_map)

    def _one_iter(self):
        values = list(self.idx_map.values())
        return iter(random.sample(values, len(values)) if self._shuffle else values)

    def __iter__(self):
        if not self._infinite:
            return self._one_iter()

        def forever():
            while True:
                yield from self._one_iter()
        return forever()


class ScanBatchLoader:
    """evaluation loader: ``((image, target), filenames, (partitions, scans))`` per scan through the `val` transform
    (``CenterCrop``, semi_seg/augment.py:35-37); ``len()`` = number of scans (what ``EvalEpocher`` reads)"""

    def __init__(self, store, out_hw=(224, 224), batch_sampler=None):
        if store.targets is None:
            raise ValueError("ScanBatchLoader: an evaluation store needs label maps")
        self.dataset = store
        self.batch_sampler = batch_sampler or ScanBatchSampler(store, shuffle=False)
        self._views = RecipeViews(store.images, dict(degrees=0.0, flips=False, pad=0, crop_first=True, brightness=None,
                                                     contrast=None, resize=None), out_hw, labels=store.targets)

    def __len__(self):
        return len(self.batch_sampler)

    def __iter__(self):
        for idx in self.batch_sampler:
            img, tgt = self._views.val(idx)
            metas = [self.dataset.meta(i) for i in idx]
            yield (img, tgt), [m[0] for m in metas], ([m[1] for m in metas], [m[2] for m in metas])


class UnlabeledDeviceLoader:
    """the unlabelled loader of ``get_data``: pre-training only takes its ``.dataset`` (semi_seg/trainers/_helper.py:31-36),
    fine-tuning never reads it (``FineTuneEpocher``); iterating gives the data set's pre-train views without label maps"""

    def __init__(self, store, *, batch_size, shuffle=True, out_hw=(224, 224), **_ignored):
        self.dataset, self._batch_size = store, int(batch_size)
        self._sampler = InfiniteRandomSampler(store, shuffle=shuffle)
        self._out_hw, self._views, self._it = tuple(out_hw), None, None

    def __iter__(self):
        self._it = iter(self._sampler)
        return self

    def __next__(self):
        if self._it is None:
            self._it = iter(self._sampler)
        if self._views is None:
            name = "prostate_pretrain" if getattr(self.dataset, "data_name", "acdc") == "prostate" else "acdc_pretrain"
            self._views = RecipeViews(self.dataset.images, name, self._out_hw)
        idx = [next(self._it) for _ in range(self._batch_size)]
        img, img_tf = self._views(idx)
        metas = [self.dataset.meta(i) for i in idx]
        tgt = torch.zeros(len(idx), 1, 1, 1, dtype=torch.long, device=img.device)
        return (img, img_tf, tgt, tgt), [m[0] for m in metas], ([m[1] for m in metas], [m[2] for m in metas])


def create_infinite_loader(dataset, shuffle=True, num_workers: int = 8, batch_size: int = 4, out_hw=(224, 224)):
    """creator.py:87-94.  ``num_workers`` is accepted and ignored (no worker processes).  The training transform of BOTH
    modes is the data set's `pretrain` recipe (creator.py:30: ``tra_transform = aug_transform.pretrain``), image and label
    map through one geometry; ``total_freedom`` decides whether the second returned pair is an independent draw."""
    if dataset.targets is None:
        return UnlabeledDeviceLoader(dataset, batch_size=batch_size, shuffle=shuffle, out_hw=out_hw)
    name = "prostate_pretrain" if getattr(dataset, "data_name", "acdc") == "prostate" else "acdc_pretrain"
    return LabeledDeviceLoader(dataset, batch_size=batch_size, sampler=InfiniteRandomSampler(dataset, shuffle=shuffle),
                               out_hw=out_hw, recipe=name, total_freedom=getattr(dataset, "total_freedom", True))


def get_data_loaders(data_params, labeled_loader_params, unlabeled_loader_params, pretrain=False, group_test=True,
                     total_freedom=False, load_predefined_list=True):
    data_name = data_params["name"]
    out_hw = _OUT_HW.get(data_name, (224, 224))
    tra_set, test_set = create_dataset(data_name, total_freedom)
    if len(tra_set.get_scan_list()) == 0 or len(test_set.get_scan_list()) == 0:
        raise RuntimeError("dataset error")
    train_scan_num = len(tra_set.get_scan_list())
    labeled_scan_num = data_params["labeled_scan_num"]
    if labeled_scan_num > train_scan_num:
        raise RuntimeError(f"labeled scan number {labeled_scan_num} greater than the train set size: {train_scan_num}")
    labeled_data_ratio = float(labeled_scan_num / train_scan_num)
    if pretrain:
        labeled_data_ratio = 0.5
        label_set, unlabeled_set = split_dataset(tra_set, labeled_data_ratio)
    elif load_predefined_list and labeled_data_ratio < 1:
        label_set, unlabeled_set = split_dataset_with_predefined_filenames(tra_set, data_name,
                                                                           labeled_ratio=labeled_data_ratio)
    else:
        label_set, unlabeled_set = split_dataset(tra_set, labeled_data_ratio)
    if len(label_set.get_scan_list()) == 0:
        raise RuntimeError("void labeled dataset, split dataset error")
    labeled_loader = create_infinite_loader(label_set, out_hw=out_hw, **labeled_loader_params)
    unlabeled_loader = create_infinite_loader(unlabeled_set, out_hw=out_hw, **unlabeled_loader_params) \
        if len(unlabeled_set) else UnlabeledDeviceLoader(unlabeled_set, batch_size=1, out_hw=out_hw)
    test_loader = ScanBatchLoader(test_set, out_hw=out_hw)  # (group_test: the two mirrored data sets are scan-grouped)
    return labeled_loader, unlabeled_loader, test_loader


def create_val_loader(*, test_loader):
    test_dataset = test_loader.dataset
    val_set, test_set = split_dataset(test_dataset, 0.35)
    out_hw = test_loader._views.out_hw
    return ScanBatchLoader(val_set, out_hw=out_hw), ScanBatchLoader(test_set, out_hw=out_hw)


def get_data(data_params, labeled_loader_params, unlabeled_loader_params, pretrain=False, total_freedom=False):
    with _seeded(1):  # @fix_seed
        labeled_loader, unlabeled_loader, test_loader = get_data_loaders(
            data_params=data_params, labeled_loader_params=labeled_loader_params,
            unlabeled_loader_params=unlabeled_loader_params, pretrain=pretrain, group_test=True,
            total_freedom=total_freedom)
        val_loader, test_loader = create_val_loader(test_loader=test_loader)
    return labeled_loader, unlabeled_loader, val_loader, test_loader
