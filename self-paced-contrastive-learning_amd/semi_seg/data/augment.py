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


def _upload_i32(rows, device):
    """parameter rows (nested lists of ints) -> int32 device tensor WITHOUT a host-device memcpy: the bytes travel as kernel
    arguments (``spcl_stage_bytes``, 3.5 KB per launch).  ``torch.tensor(rows).to(device)`` copies from pageable memory,
    which makes the host wait until the stream has drained -- once per batch: the loader could never run ahead of the step
    (measured: 1.19 ms per pre-train step of 30 slices on the product's data path, 0.98 - 1.0 with this upload;
    tools/diag/pretrain_epoch_time.py ... real)."""
    import ctypes

    import numpy as np
    arr = np.ascontiguousarray(np.asarray(rows, dtype=np.int32))
    out = torch.empty(arr.shape, dtype=torch.int32, device=device)
    if arr.size:
        _n.call("spcl_stage_bytes", ctypes.c_void_p(out.data_ptr()), arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes,
                _n.stream())
    return out


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


# ---- round 5: every recipe of semi_seg/augment.py, with the interpolation the reference's wrapper selects
# (contrastyou/augment/synchronize.py:95-103: BILINEAR on images, NEAREST on targets) -- spcl_augment_views_recipe
RECIPE_W = 28
RECIPES = {
    # ACDCStrongTransforms.pretrain (semi_seg/augment.py:6-22)
    "acdc_pretrain": dict(degrees=45.0, flips=True, pad=0, crop_first=False, brightness=(0.5, 1.5), contrast=(0.5, 1.5), resize=None),
    # ProstateStrongTransforms.pretrain (:54-69): Resize(224) happens once, when the store is built (``resize_store``)
    "prostate_pretrain": dict(degrees=10.0, flips=True, pad=20, crop_first=False, brightness=(0.9, 1.1), contrast=(0.9, 1.1),
                              resize=224),
    # ACDCStrongTransforms.label (:23-34): RandomCrop(224), THEN RandomRotation(30) of the crop; no colour jitter
    "acdc_label": dict(degrees=30.0, flips=False, pad=0, crop_first=True, brightness=None, contrast=None, resize=None),
    # ProstateStrongTransforms.label (:70-80): Resize(224), RandomCrop(224)
    "prostate_label": dict(degrees=0.0, flips=False, pad=0, crop_first=True, brightness=None, contrast=None, resize=224),
}


def pil_rotate_matrix(angle: float, ws: int, hs: int):
    """the six doubles ``Image.rotate(angle)`` hands its transform (PIL/Image.py): what the bilinear rotation works with"""
    a = -math.radians(angle % 360.0)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    cx, cy = ws / 2, hs / 2
    m[2] = m[0] * (-cx) + m[1] * (-cy) + m[2]
    m[5] = m[3] * (-cx) + m[4] * (-cy) + m[5]
    m[2] += cx
    m[5] += cy
    return m


def _f64_words(x: float):
    lo, hi = struct.unpack("<ii", struct.pack("<d", x))
    return [lo, hi]


def recipe_row(slice_index, src_hw, out_hw, *, angle=0.0, vflip=False, hflip=False, top=0, left=0, pad=0, crop_first=False,
               brightness=1.0, contrast=1.0, contrast_first=False, bilinear=True):
    """one parameter row of ``spcl_augment_views_recipe`` (28 ints) from the drawn values"""
    (hs, ws), (oh, ow) = src_hw, out_hw
    rw, rh = (ow, oh) if crop_first else (ws, hs)  # the image the rotation acts on
    flags = (1 if hflip else 0) | (2 if vflip else 0) | (4 if contrast_first else 0) | (8 if bilinear else 0) | (16 if crop_first else 0)
    row = [slice_index, flags, int(top), int(left), int(pad), _f32_bits(float(brightness)), _f32_bits(float(contrast)), 0]
    row += pil_affine_q16(angle, rw, rh)
    for v in pil_rotate_matrix(angle, rw, rh):
        row += _f64_words(v)
    row += [0] * (RECIPE_W - len(row))
    return row


def draw_recipe_params(slice_index, src_hw, out_hw, recipe, rng=random):
    """the random draws of one view in the order torchvision makes them for the recipe's common transform, then the jitter"""
    (hs, ws), (oh, ow) = src_hw, out_hw
    r = recipe
    pad = int(r["pad"])

    def crop():
        return rng.randint(0, hs + 2 * pad - oh), rng.randint(0, ws + 2 * pad - ow)
    if r["crop_first"]:
        top, left = crop()
        angle = rng.uniform(-r["degrees"], r["degrees"]) if r["degrees"] else 0.0
        vflip = hflip = False
    else:
        angle = rng.uniform(-r["degrees"], r["degrees"]) if r["degrees"] else 0.0
        vflip = bool(r["flips"] and rng.random() < 0.5)
        hflip = bool(r["flips"] and rng.random() < 0.5)
        top, left = crop()
    b = rng.uniform(*r["brightness"]) if r["brightness"] else 1.0
    c = rng.uniform(*r["contrast"]) if r["contrast"] else 1.0
    cf = bool(r["brightness"] and rng.random() < 0.5)
    return recipe_row(slice_index, src_hw, out_hw, angle=angle, vflip=vflip, hflip=hflip, top=top, left=left, pad=pad,
                      crop_first=r["crop_first"], brightness=b, contrast=c, contrast_first=cf, bilinear=True)


def redraw_jitter(row, recipe, rng=random):
    """a second version of a view under ``SequentialWrapperTwice(total_freedom=False)`` (contrastyou/augment/synchronize.py:
    129-150): the common transform replays the same seed -- same geometry --, the image transform draws again"""
    r = list(row)
    b = rng.uniform(*recipe["brightness"]) if recipe["brightness"] else 1.0
    c = rng.uniform(*recipe["contrast"]) if recipe["contrast"] else 1.0
    cf = bool(recipe["brightness"] and rng.random() < 0.5)
    r[1] = (r[1] & ~4) | (4 if cf else 0)
    r[5], r[6] = _f32_bits(float(b)), _f32_bits(float(c))
    return r


def center_crop_row(slice_index, src_hw, out_hw):
    """``CenterCrop`` (the `val` transform, semi_seg/augment.py:35-37): torchvision's int(round((h - oh) / 2.0)) offsets"""
    (hs, ws), (oh, ow) = src_hw, out_hw
    return recipe_row(slice_index, src_hw, out_hw, top=int(round((hs - oh) / 2.0)), left=int(round((ws - ow) / 2.0)),
                      crop_first=True, bilinear=False)


def resize_coeffs(in_size: int, out_size: int):
    """Resample.c ``precompute_coeffs`` + ``normalize_coeffs_8bpc`` for the bilinear filter -> (bounds [out][2], kk [out][ksize])"""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 1.0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds, kk = [], []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / fscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = []
        for x in range(xmax):
            t = (x + xmin - center + 0.5) * ss
            t = -t if t < 0.0 else t
            w.append(1.0 - t if t < 1.0 else 0.0)
        ww = 0.0
        for v in w:
            ww += v
        row = [0] * ksize
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            row[x] = int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22))
        bounds.append([xmin, xmax])
        kk.append(row)
    return bounds, kk, ksize


def resize_shorter_edge(hw, size: int):
    """torchvision ``Resize(int)``: the shorter edge becomes ``size``, the other keeps the aspect ratio"""
    h, w = hw
    if (w <= h and w == size) or (h <= w and h == size):
        return h, w
    if w < h:
        return int(size * h / w), size
    return size, int(size * w / h)


def resize_store(images: torch.Tensor, size: int) -> torch.Tensor:
    """``transforms.Resize(size)`` (bilinear, PIL's arithmetic) of every slice of an 8-bit store [S, H, W] -> [S, oh, ow]"""
    _n.require_gpu(images)
    S, HS, WS = images.shape
    oh, ow = resize_shorter_edge((HS, WS), size)
    if (oh, ow) == (HS, WS):
        return images
    bx, kx, ksx = resize_coeffs(WS, ow)
    by, ky, ksy = resize_coeffs(HS, oh)
    dev = images.device
    t = lambda a: torch.tensor(a, dtype=torch.int32).to(dev)  # noqa: E731
    tbx, tkx, tby, tky = t(bx), t(kx), t(by), t(ky)
    tmp = torch.empty(S, HS, ow, dtype=torch.float32, device=dev)
    out = torch.empty(S, oh, ow, dtype=torch.float32, device=dev)
    src = images.float().contiguous()
    _n.call("spcl_resize_bilinear_pil", _n.ptr(src), S, HS, WS, _n.ptr(tbx), _n.ptr(tkx), ksx, _n.ptr(tby), _n.ptr(tky), ksy,
            _n.ptr(tmp), _n.ptr(out), oh, ow, _n.stream())
    return out


class RecipeViews:
    """views of a device store by one of ``RECIPES`` (or a dict of the same keys): ``pairs(indices)`` -> two INDEPENDENT
    views per slice (SequentialWrapperTwice(total_freedom=True): the pre-train recipes); ``labelled(indices)`` -> ONE geometry
    per slice applied to image (bilinear) and label map (nearest) -- the `label` recipes, whose two returned versions share
    the common seed and have no image-only randomness: they are the same tensors twice."""

    def __init__(self, images: torch.Tensor, recipe="acdc_pretrain", out_hw=(224, 224), labels: torch.Tensor = None):
        _n.require_gpu(images)
        self.recipe = dict(RECIPES[recipe]) if isinstance(recipe, str) else dict(recipe)
        if self.recipe.get("resize"):
            images = resize_store(images, int(self.recipe["resize"]))
            if labels is not None:
                raise NotImplementedError("label maps are resized with NEAREST by the reference: resize them before building the store")
        self.images, self.out_hw = images.float().contiguous(), tuple(out_hw)
        self.labels = labels.to(torch.uint8).contiguous() if labels is not None else None

    def rows(self, indices: Sequence[int], rng=random):
        hw = tuple(self.images.shape[1:])
        return [draw_recipe_params(i, hw, self.out_hw, self.recipe, rng) for i in indices]

    def apply(self, rows, with_labels=False):
        S, HS, WS = self.images.shape
        oh, ow = self.out_hw
        dev = self.images.device
        p = _upload_i32(rows, dev)
        out = torch.empty(len(rows), 1, oh, ow, dtype=torch.float32, device=dev)
        lab_out = torch.empty(len(rows), 1, oh, ow, dtype=torch.int64, device=dev) if with_labels else None
        # (the split form: 16 workgroups per view instead of one, every pixel sampled once -- include/spcl_hip.h)
        ws = torch.empty(_n.call("spcl_augment_views_recipe_workspace_bytes", len(rows), oh, ow) // 8 + 1, dtype=torch.int64,
                         device=dev)
        _n.call("spcl_augment_views_recipe_ws", _n.ptr(self.images), _n.ptr(self.labels) if with_labels else None, S, HS, WS,
                _n.ptr(p), len(rows), _n.ptr(out), _n.ptr(lab_out), oh, ow, int(self.recipe["pad"]), _n.ptr(ws),
                ws.numel() * 8, _n.stream())
        return (out, lab_out) if with_labels else out

    def pairs(self, indices: Sequence[int], rng=random):
        both = self.apply(self.rows(indices, rng) + self.rows(indices, rng))
        n = len(indices)
        return both[:n], both[n:]

    __call__ = pairs

    def labelled(self, indices: Sequence[int], rng=random):
        assert self.labels is not None, "RecipeViews.labelled needs the store's label maps"
        return self.apply(self.rows(indices, rng), with_labels=True)

    def val(self, indices: Sequence[int]):
        """``CenterCrop(out_hw)`` of image (and label map when the store has one): the reference's `val` transform"""
        hw = tuple(self.images.shape[1:])
        rows = [center_crop_row(i, hw, self.out_hw) for i in indices]
        return self.apply(rows, with_labels=self.labels is not None)


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
        width = len(rows[0])
        if width not in (8, 12) or any(len(r) != width for r in rows):
            raise ValueError("PretrainViews.apply: parameter rows of 12 ints (PIL-exact) or 8 ints (float recipe)")
        p = _upload_i32(rows, self.images.device)
        out = torch.empty(len(rows), 1, oh, ow, dtype=torch.float32, device=self.images.device)
        entry = "spcl_augment_views_pil" if width == 12 else "spcl_augment_views"
        _n.call(entry, _n.ptr(self.images), S, HS, WS, _n.ptr(p), len(rows), _n.ptr(out), oh, ow, _n.stream())
        return out

    def __call__(self, indices: Sequence[int], rng=random):
        both = self.apply(self.params(indices, rng))
        n = len(indices)
        return both[:n], both[n:]
