"""The supervised criterion of the fine-tune / evaluation path: ``deepclustering2.loss.KL_div`` and
``deepclustering2.utils.class2one_hot`` as the reference calls them (``val.py:9``, ``main_pretrain_encoder.py:56``,
``semi_seg/epochers/new_epocher.py:84-86,270-271``).  deepclustering2 is an un-vendored, unpinned third party (SURVEY
8c): restated from its published definition, arithmetic in HIP (``spcl_kl_div_*``, ``spcl_one_hot``)."""
import torch
from torch import nn

from ... import functional as F_hip


def simplex(t: torch.Tensor, axis=1) -> bool:
    s = t.sum(axis).float()
    return bool(torch.allclose(s, torch.ones_like(s), rtol=1e-4, atol=1e-4))


def class2one_hot(seg: torch.Tensor, C: int) -> torch.Tensor:
    """[B,H,W] integer labels -> [B,C,H,W] one-hot (float32; the reference's int one-hot is only ever multiplied)."""
    if seg.dim() == 2:
        seg = seg.unsqueeze(0)
    assert seg.dim() == 3, seg.shape
    return F_hip.one_hot_classes(seg, C)


class KL_div(nn.Module):
    """``KL_div(reduction='mean', eps=1e-16)(prob, target)`` = mean over batch and positions of
    ``sum_c -target * log((prob + eps) / (target + eps))``; asserts both inputs are simplexes unless
    ``disable_assert=True`` (one device->host sync, like the reference)."""

    def __init__(self, reduction="mean", eps=1e-16, weight=None, verbose=True):
        super().__init__()
        if reduction != "mean" or weight is not None:
            raise NotImplementedError("KL_div mirror: reduction='mean' without class weights (what the reference uses)")
        self._eps = eps

    def forward(self, prob: torch.Tensor, target: torch.Tensor, **kwargs) -> torch.Tensor:
        if not kwargs.get("disable_assert"):
            assert prob.shape == target.shape
            assert simplex(prob), "prob is not a simplex"
            assert simplex(target), "target is not a simplex"
        return F_hip.kl_div(prob, target, self._eps)
