"""Device-resident slice stores with the reference's meta-labels (``semi_seg/data/dataset.py:16-71`` on top of
``contrastyou/data/dataset/base.py:76-203``).

The reference keeps PNG folders and decodes / augments them with PIL in DataLoader workers; here ALL slices of the
training scans live in HBM as one ``[S, H, W]`` f32 tensor in [0, 1] (ACDC: ~1 900 slices of 256x256 = 0.5 GB of the 288
GB) and a batch is gathered and augmented by one kernel launch (``augment.PretrainViews``).  A slice is identified by
its file stem ``<scan>_<slice index>`` (``patient004_00_07``); scan name and partition are derived exactly as the
reference does (``group_re`` search, ``_get_partition`` arithmetic on the LAST number of the stem)."""
import os
import re
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence

import torch

from .rearr import ContrastDataset


def acdc_partition(filename: str, scan_len: int, partition_num: int = 3) -> str:
    """``ACDCDataset._get_partition`` (semi_seg/data/dataset.py:34-43): thirds of the scan by slice index"""
    cutting_point = scan_len // partition_num
    cur_index = int(re.compile(r"\d+").findall(filename)[-1])
    if cur_index <= cutting_point - 1:
        return str(0)
    if cur_index <= 2 * cutting_point:
        return str(1)
    return str(2)


def prostate_partition(filename: str, scan_len: int, partition_num: int = 8) -> str:
    """``ProstateDataset._get_partition`` (semi_seg/data/dataset.py:66-71)"""
    cutting_point = scan_len // partition_num
    cur_index = int(re.compile(r"\d+").findall(filename)[-1])
    return str(cur_index // (cutting_point + 1))


class DeviceSliceStore(ContrastDataset):
    """``images`` [S, H, W] f32 in [0,1] on the device, ``filenames`` the S stems; ``scan_info`` scan -> number of
    slices (the reference's ``acdc_info.npy`` / ``prostate_info.npy``), derived from the stems when not given."""
    partition_num = 3
    group_re = r"patient\d+_\d+"
    data_name = "acdc"

    def __init__(self, images: torch.Tensor, filenames: Sequence[str], scan_info: Optional[Dict[str, int]] = None,
                 targets: Optional[torch.Tensor] = None):
        assert images.dim() == 3 and images.shape[0] == len(filenames), (images.shape, len(filenames))
        self.images = images.float().contiguous()
        self.targets = targets
        self._filenames = [os.path.splitext(os.path.basename(f))[0] for f in filenames]
        self._re = re.compile(self.group_re)
        if scan_info is None:
            scan_info = {}
            for f in self._filenames:
                s = self._get_scan_name(f)
                scan_info[s] = max(scan_info.get(s, 0), int(re.findall(r"\d+", f)[-1]) + 1)
        self._scan_info = dict(scan_info)
        self._name = f"{type(self).__name__}-train"
        self._is_preload = True

    # ---- the interface ContrastBatchSampler and the label generators use
    def get_memory_dictionary(self):
        return OrderedDict(img=list(self._filenames))

    def _get_scan_name(self, filename=None, stem=None) -> str:
        m = self._re.search(stem if stem is not None else filename)
        if m is None:
            raise AttributeError(f"Cannot match pattern: {self.group_re} for {filename or stem}")
        return m.group(0)

    def _get_partition(self, filename) -> str:
        return acdc_partition(filename, self._scan_info[self._get_scan_name(filename)], self.partition_num)

    def show_partitions(self) -> List[str]:
        return [self._get_partition(f) for f in self._filenames]

    def show_scan_names(self) -> List[str]:
        return [self._get_scan_name(f) for f in self._filenames]

    def get_scan_list(self):
        return sorted(set(self.show_scan_names()))

    def __len__(self):
        return len(self._filenames)

    def meta(self, index):
        f = self._filenames[index]
        return f, self._get_partition(f), self._get_scan_name(f)

    @classmethod
    def from_folder(cls, root: str, device="cuda", size: Optional[int] = None):
        """PNG folder ``root/img/*.png`` (+ ``root/gt/*.png`` label maps of the same stems when present: the reference's layout,
        contrastyou/data/dataset/base.py:76-140 ``sub_folders``) -> store.  Slices are centre-padded / cropped to ONE size (the
        store is one tensor); grey levels are kept as k / 255, label maps as uint8 class codes.  ``<name>_info.npy`` (scan ->
        number of slices) is read when it lies next to the folders (semi_seg/data/dataset.py:28-31)."""
        from PIL import Image
        import numpy as np
        files = sorted(f for f in os.listdir(os.path.join(root, "img")) if f.lower().endswith(".png"))
        arrs = [np.asarray(Image.open(os.path.join(root, "img", f)).convert("L"), dtype=np.uint8) for f in files]
        gt_dir = os.path.join(root, "gt")
        gts = None
        if os.path.isdir(gt_dir):
            missing = [f for f in files if not os.path.exists(os.path.join(gt_dir, f))]
            if missing:
                raise FileNotFoundError(f"label maps missing for {len(missing)} slices, e.g. {missing[0]}")
            gts = [np.asarray(Image.open(os.path.join(gt_dir, f)).convert("L"), dtype=np.uint8) for f in files]
            assert all(a.shape == g.shape for a, g in zip(arrs, gts)), "a slice and its label map differ in size"
        size = size or max(max(a.shape) for a in arrs)

        def centred(list_of_arrays, dtype):
            out = torch.zeros(len(list_of_arrays), size, size, dtype=dtype)
            for k, a in enumerate(list_of_arrays):
                h, w = a.shape
                t = torch.from_numpy(np.ascontiguousarray(a))[max(0, (h - size) // 2):max(0, (h - size) // 2) + size,
                                                              max(0, (w - size) // 2):max(0, (w - size) // 2) + size]
                oy, ox = (size - t.shape[0]) // 2, (size - t.shape[1]) // 2
                out[k, oy:oy + t.shape[0], ox:ox + t.shape[1]] = t.to(dtype)
            return out
        images = centred(arrs, torch.float32) / 255.0
        targets = centred(gts, torch.uint8) if gts is not None else None
        info = None
        for name in ("acdc_info.npy", "prostate_info.npy"):
            p = os.path.join(root, name)
            if os.path.exists(p):
                info = np.load(p, allow_pickle=True).item()
        return cls(images.to(device), files, info, targets=targets.to(device) if targets is not None else None)


class ACDCSliceStore(DeviceSliceStore):
    pass


class ProstateSliceStore(DeviceSliceStore):
    partition_num = 8
    group_re = r"Case\d+"
    data_name = "prostate"

    def _get_partition(self, filename) -> str:
        return prostate_partition(filename, self._scan_info[self._get_scan_name(filename)], self.partition_num)


def synthetic_slice_store(scans=12, slices_per_scan=(8, 11), size=256, device="cuda", seed=0, kind="acdc"):
    """ACDC- / Prostate-shaped synthetic store (there is no data set in the build image): smooth random blobs, scans of
    varying length, stems ``patientNNN_00_SS`` / ``CaseNN_SS``."""
    g = torch.Generator().manual_seed(seed)
    names, imgs = [], []
    lo, hi = slices_per_scan
    for s in range(scans):
        n = lo + int(torch.randint(0, hi - lo + 1, (1,), generator=g))
        base = torch.nn.functional.interpolate(torch.rand(1, 1, 8, 8, generator=g), size=(size, size), mode="bilinear",
                                               align_corners=False)[0, 0]
        for k in range(n):
            imgs.append((base * (0.6 + 0.4 * k / n) + 0.1 * torch.rand(size, size, generator=g)).clamp_(0, 1))
            names.append(f"patient{s + 1:03d}_00_{k:02d}" if kind == "acdc" else f"Case{s:02d}_{k:02d}")
    cls = ACDCSliceStore if kind == "acdc" else ProstateSliceStore
    return cls(torch.stack(imgs).to(device), names)
