"""Label generators of ``semi_seg/epochers/helper.py:48-65`` (sklearn LabelEncoder == rank among sorted uniques) and
the seeded per-sample flip the epocher and the hook share (deepclustering2 ``TensorRandomFlip(axis=[1,2],
threshold=0.8)`` under ``FixRandomSeed(seed)``; un-vendored third party, restated by contract -- SURVEY 8c)."""
import random
from typing import List

import torch


def _label_encode(items):
    index = {v: i for i, v in enumerate(sorted(set(items)))}
    return [index[v] for v in items]


class PartitionLabelGenerator:
    def __call__(self, partition_list: List[str], **kwargs):
        return _label_encode(list(partition_list))


class PatientLabelGenerator:
    def __call__(self, patient_list: List[str], **kwargs):
        return _label_encode(list(patient_list))


class ACDCCycleGenerator:
    def __call__(self, experiment_list: List[str], **kwargs):
        return [0 if e == "00" else 1 for e in experiment_list]


class SIMCLRGenerator:
    def __call__(self, partition_list: List[str], **kwargs):
        return list(range(len(partition_list)))


class FixRandomSeed:
    """Context manager: seed python's and numpy's RNGs, restore their states on exit (contract of
    deepclustering2.FixRandomSeed as used at semi_seg/epochers/new_pretrain.py:57,64 and semi_seg/hooks/infonce.py:177,
    211-214 -- the dense hook draws its points with ``np.random.choice``)."""

    def __init__(self, seed):
        self._seed = seed

    def __enter__(self):
        import numpy as np
        self._state, self._np_state = random.getstate(), np.random.get_state()
        random.seed(self._seed)
        np.random.seed(self._seed % (2 ** 32))
        return self

    def __exit__(self, *a):
        import numpy as np
        random.setstate(self._state)
        np.random.set_state(self._np_state)


class TensorRandomFlip:
    """Flip a [C,H,W] sample along each of ``axis`` independently when u < threshold (u ~ python ``random``)."""

    def __init__(self, axis=(1, 2), threshold=0.8):
        self._axis = tuple(axis)
        self._threshold = threshold
        self._plans = {}

    def decisions(self, n):
        return [[random.random() < self._threshold for _ in self._axis] for _ in range(n)]

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        dims = [a for a in self._axis if random.random() < self._threshold]
        return x.flip(dims) if dims else x

    def apply_pair(self, first: torch.Tensor, second: torch.Tensor) -> torch.Tensor:
        """``torch.cat([first, stack([self(s) for s in second])])`` (new_pretrain.py:57-58,93) as ONE HIP launch
        (``spcl_flip_pair``) for contiguous [N,C,H,W] GPU tensors of one shape and dtype; draws the same random stream as
        ``apply_batch(second)``."""
        if not (first.is_cuda and second.is_cuda and first.dim() == 4 and first.shape == second.shape
                and first.dtype == second.dtype and tuple(self._axis) == (1, 2)
                and not first.requires_grad and not second.requires_grad):
            return torch.cat([first, self.apply_batch(second)], dim=0)
        from ... import native as _n
        dec = tuple(tuple(d) for d in self.decisions(second.shape[0]))
        key = (dec, second.device, "flags")
        flags = self._plans.get(key)
        if flags is None:
            flags = torch.tensor([int(d[0]) | (int(d[1]) << 1) for d in dec], dtype=torch.uint8, device=second.device)
            if len(self._plans) < 256:
                self._plans[key] = flags
        a, b = first.contiguous(), second.contiguous()
        N, C, H, W = a.shape
        out = torch.empty((2 * N, C, H, W), dtype=a.dtype, device=a.device)
        _n.call("spcl_flip_pair", _n.ptr(a), _n.ptr(b), _n.ptr(out), a.element_size(), N, C, H, W, _n.ptr(flags),
                _n.stream())
        return out

    def apply_batch(self, x: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        """Batched equivalent of ``stack([self(s) for s in x])`` drawing the same random stream.

        GPU tensors: ONE HIP launch (``spcl_flip_batch``) driven by a per-sample flag byte (bit 0: flip H, bit 1:
        flip W); the flag tensor is cached per decision pattern, so a repeated pattern costs no host->device copy and
        the call can be captured in a hipGraph.  ``out`` (same shape, contiguous) receives the result when given.
        CPU tensors (host-side tests of the random stream only): one gather per flip pattern with torch ops."""
        dec = tuple(tuple(d) for d in self.decisions(x.shape[0]))
        if x.is_cuda:
            if x.dim() != 4 or tuple(self._axis) != (1, 2):
                raise NotImplementedError("the HIP flip handles [N,C,H,W] batches flipped along H and/or W")
            from ... import native as _n
            key = (dec, x.device, "flags")
            flags = self._plans.get(key)
            if flags is None:
                flags = torch.tensor([int(d[0]) | (int(d[1]) << 1) for d in dec], dtype=torch.uint8, device=x.device)
                if len(self._plans) < 256:
                    self._plans[key] = flags
            if x.requires_grad and out is None:  # features on the autograd graph (the dense hook): differentiable form
                from ... import functional as _F
                return _F.flip_batch(x, flags)
            xc = x.contiguous()
            if out is None:
                out = torch.empty_like(xc)
            elif out.shape != xc.shape or not out.is_contiguous() or out.dtype != xc.dtype:
                raise ValueError("apply_batch: `out` must be a contiguous tensor of x's shape and dtype")
            N, C, H, W = xc.shape
            _n.call("spcl_flip_batch", _n.ptr(xc), _n.ptr(out), xc.element_size(), N, C, H, W, _n.ptr(flags),
                    _n.stream())
            return out
        key = (dec, x.device)
        plan = self._plans.get(key)
        if plan is None:
            groups = {}
            for i, d in enumerate(dec):
                groups.setdefault(d, []).append(i)
            plan = [([a + 1 for a, f in zip(self._axis, pat) if f],
                     torch.tensor(idx, dtype=torch.long, device=x.device)) for pat, idx in groups.items() if any(pat)]
            if len(self._plans) < 256:
                self._plans[key] = plan
        res = x.clone() if out is None else out.copy_(x)
        for dims, idx in plan:
            res.index_copy_(0, idx, x.index_select(0, idx).flip(dims))
        return res
