"""Mirror of the pieces of ``contrastyou/projectors/nn.py`` the encoder projector needs (:8-15 Flatten,
:29-36 Normalize, :39-45 Identical, :56-63 pool factory, :66-88 _ProjectorHeadBase).  The modules exist so that
``_header`` keeps the reference's Sequential indices (state_dict keys ``_header.2.*`` / ``_header.4.*``);
ProjectionHead.forward does not iterate them -- it calls the fused HIP projector."""
from typing import Tuple

from torch import nn
from torch.nn.modules.utils import _pair


class Flatten(nn.Module):
    def forward(self, features):
        return features.view(features.shape[0], -1)


class Normalize(nn.Module):
    def __init__(self, dim=1) -> None:
        super().__init__()
        self._dim = dim

    def forward(self, input):
        return nn.functional.normalize(input, p=2, dim=self._dim)


class Identical(nn.Module):
    def forward(self, input):
        return input


def _check_head_type(head_type):
    return head_type in ("mlp", "linear")


def _check_pool_name(pool_name):
    return pool_name in ("adaptive_avg", "adaptive_max", "identical", "none")


def get_pool_component(pool_name, spatial_size: Tuple[int, int]):
    return {"adaptive_avg": nn.AdaptiveAvgPool2d(spatial_size), "adaptive_max": nn.AdaptiveMaxPool2d(spatial_size),
            None: Identical(), "none": Identical(), "identical": Identical()}[pool_name]


class _ProjectorHeadBase(nn.Module):
    def __init__(self, *, input_dim: int, output_dim: int, head_type: str, normalize: bool, pool_name="adaptive_avg",
                 spatial_size=(1, 1)):
        super().__init__()
        self._input_dim = input_dim
        self._output_dim = output_dim
        assert _check_head_type(head_type=head_type)
        self._head_type = head_type
        self._normalize = normalize
        assert _check_pool_name(pool_name=pool_name)
        self._pool_name = pool_name
        self._spatial_size = _pair(spatial_size)
        self._pooling_module = get_pool_component(self._pool_name, self._spatial_size)
