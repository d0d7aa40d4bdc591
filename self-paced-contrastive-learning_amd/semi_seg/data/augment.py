"""The pre-train augmentation recipe on device (``semi_seg/augment.py:6-22`` ``ACDCStrongTransforms.pretrain``,
``SequentialWrapperTwice(total_freedom=True)``: two INDEPENDENT views per slice): RandomRotation(45) ->
RandomVerticalFlip -> RandomHorizontalFlip -> RandomCrop(224) -> ColorJitter(brightness [0.5,1.5], contrast [0.5,1.5])
-> ToTensor, parameters drawn on the host from python's ``random`` like torchvision does from torch's, geometry and
colour applied by ONE HIP launch per batch (``spcl_augment_views``, csrc/augment.hip) reading the device-resident slice
store.  Saturation / hue of the jitter are the identity on one-channel images.  PIL quantises to 8 bits between the
steps, and so does the default path here (``pil_exact=True``: ``spcl_augment_views_pil`` restates PIL's fixed-point nearest
rotation and its truncating 8-bit blends, bit-exact against views PIL itself produced: tests/golden/g9_augment.npz,
oracle.augment_view_pil); ``pil_exact=False`` keeps the round-2 float recipe (``spcl_augment_views``: any float store, no
8-bit quantisation between the steps, pinned by oracle.augment_view)."""
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


def pil_affine_q16(angle: float, ws: int, hs: int):
    """the six 16.16 coefficients ``Image.rotate(angle, NEAREST)`` uses (PIL/Image.py rotate + Geometry.c affine_fixed:
    double-precision matrix about (w/2, h/2), entries round(.., 15), FIX(v) = floor(v * 65536 + 0.5))"""
    a = -math.radians(angle % 360.0)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    cx, cy = ws / 2, hs / 2
    m[2] = m[0] * (-cx) + m[1] * (-cy) + m[2]
    m[5] = m[3] * (-cx) + m[4] * (-cy) + m[5]
    m[2] += cx
    m[5] += cy

    def fix(v):
        return int(math.floor(v * 65536.0 + 0.5))
    return [fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]),
            fix(m[5] + m[3] * 0.5 + m[4] * 0.5)]


def draw_view_params_pil(slice_index: int, src_hw, out_hw, *, degrees=45.0, brightness=(0.5, 1.5), contrast=(0.5, 1.5),
                         flips=True, rng=random):
    """one row of ``spcl_augment_views_pil`` parameters: [slice, a0 .. a5, flags, top, left, brightness, contrast], drawn in
    the order torchvision draws them (rotation angle, vertical flip, horizontal flip, crop, jitter factors, jitter order)"""
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
    return [slice_index] + pil_affine_q16(angle, ws, hs) + [flags, top, left, _f32_bits(b), _f32_bits(c)]


class PretrainViews:
    """``images`` [S,H,W] device store -> two views [B,1,oh,ow] each for a list of slice indices"""

    def __init__(self, images: torch.Tensor, out_hw=(224, 224), pil_exact: bool = True, **recipe):
        _n.require_gpu(images)
        self.images, self.out_hw, self.recipe, self.pil_exact = images, tuple(out_hw), recipe, bool(pil_exact)

    def params(self, indices: Sequence[int], rng=random):
        hw = tuple(self.images.shape[1:])
        draw = draw_view_params_pil if self.pil_exact else draw_view_params
        first = [draw(i, hw, self.out_hw, rng=rng, **self.recipe) for i in indices]
        second = [draw(i, hw, self.out_hw, rng=rng, **self.recipe) for i in indices]
        return first + second

    def apply(self, rows):
        """rows: parameter rows (lists of 12 ints, or 8 for the float recipe) -> [len(rows), 1, oh, ow] f32"""
        S, HS, WS = self.images.shape
        oh, ow = self.out_hw
        p = torch.tensor(rows, dtype=torch.int32).to(self.images.device, non_blocking=True)
        out = torch.empty(len(rows), 1, oh, ow, dtype=torch.float32, device=self.images.device)
        width = len(rows[0])
        if width not in (8, 12) or any(len(r) != width for r in rows):
            raise ValueError("PretrainViews.apply: parameter rows of 12 ints (PIL-exact) or 8 ints (float recipe)")
        entry = "spcl_augment_views_pil" if width == 12 else "spcl_augment_views"
        _n.call(entry, _n.ptr(self.images), S, HS, WS, _n.ptr(p), len(rows), _n.ptr(out), oh, ow, _n.stream())
        return out

    def __call__(self, indices: Sequence[int], rng=random):
        both = self.apply(self.params(indices, rng))
        n = len(indices)
        return both[:n], both[n:]
