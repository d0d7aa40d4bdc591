"""The pre-train augmentation recipe on device (``semi_seg/augment.py:6-22`` ``ACDCStrongTransforms.pretrain``,
``SequentialWrapperTwice(total_freedom=True)``: two INDEPENDENT views per slice): RandomRotation(45) ->
RandomVerticalFlip -> RandomHorizontalFlip -> RandomCrop(224) -> ColorJitter(brightness [0.5,1.5], contrast [0.5,1.5])
-> ToTensor, parameters drawn on the host from python's ``random`` like torchvision does from torch's, geometry and
colour applied by ONE HIP launch per batch (``spcl_augment_views``, csrc/augment.hip) reading the device-resident slice
store.  Saturation / hue of the jitter are the identity on one-channel images.  PIL quantises to 8 bits between the
steps; this path stays in f32 -- the distributions of the random parameters are the reference's, bit parity with PIL is
not a goal (the arithmetic is pinned by oracle.augment_view instead)."""
import math
import random
import struct
from typing import Sequence

import torch

from ... import native as _n


def _f32_bits(x: float) -> int:
    return struct.unpack("<i", struct.pack("<f", x))[0]


def draw_view_params(slice_index: int, src_hw, out_hw, *, degrees=45.0, brightness=(0.5, 1.5), contrast=(0.5, 1.5),
                     flips=True, rng=random):
    """one row of ``spcl_augment_views`` parameters: [slice, cos_q16, sin_q16, flags, top, left, brightness, contrast]"""
    (hs, ws), (oh, ow) = src_hw, out_hw
    angle = rng.uniform(-degrees, degrees) if degrees else 0.0
    flags = 0
    if flips and rng.random() < 0.5:
        flags |= 2  # RandomVerticalFlip
    if flips and rng.random() < 0.5:
        flags |= 1  # RandomHorizontalFlip
    top = rng.randint(0, hs - oh)
    left = rng.randint(0, ws - ow)
    b = rng.uniform(*brightness) if brightness else 1.0
    c = rng.uniform(*contrast) if contrast else 1.0
    if rng.random() < 0.5:
        flags |= 4  # ColorJitter applies its factors in random order
    rad = math.radians(angle)
    return [slice_index, int(round(math.cos(rad) * 65536)), int(round(math.sin(rad) * 65536)), flags, top, left,
            _f32_bits(b), _f32_bits(c)]


class PretrainViews:
    """``images`` [S,H,W] device store -> two views [B,1,oh,ow] each for a list of slice indices"""

    def __init__(self, images: torch.Tensor, out_hw=(224, 224), **recipe):
        _n.require_gpu(images)
        self.images, self.out_hw, self.recipe = images, tuple(out_hw), recipe

    def params(self, indices: Sequence[int], rng=random):
        hw = tuple(self.images.shape[1:])
        first = [draw_view_params(i, hw, self.out_hw, rng=rng, **self.recipe) for i in indices]
        second = [draw_view_params(i, hw, self.out_hw, rng=rng, **self.recipe) for i in indices]
        return first + second

    def apply(self, rows):
        """rows: parameter rows (list of 8 ints) -> [len(rows), 1, oh, ow] f32"""
        S, HS, WS = self.images.shape
        oh, ow = self.out_hw
        p = torch.tensor(rows, dtype=torch.int32).to(self.images.device, non_blocking=True)
        out = torch.empty(len(rows), 1, oh, ow, dtype=torch.float32, device=self.images.device)
        _n.call("spcl_augment_views", _n.ptr(self.images), S, HS, WS, _n.ptr(p), len(rows), _n.ptr(out), oh, ow,
                _n.stream())
        return out

    def __call__(self, indices: Sequence[int], rng=random):
        both = self.apply(self.params(indices, rng))
        n = len(indices)
        return both[:n], both[n:]
