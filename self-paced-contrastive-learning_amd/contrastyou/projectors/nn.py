"""Building blocks of the projector heads, mirror of ``contrastyou/projectors/nn.py`` (Flatten :8-15, Normalize
:29-36, Identical :39-45, pool factory :56-63, ``_ProjectorHeadBase`` :66-88).  They exist so that
``ProjectionHead._header`` keeps the reference's ``nn.Sequential`` indices -- the ``state_dict`` keys ``_header.2.*`` /
``_header.4.*`` are part of the contract -- but ``ProjectionHead.forward`` never iterates them: it calls the fused HIP
projector (functional.projector)."""
from torch import nn
from torch.nn.modules.utils import _pair

HEAD_TYPES = ("mlp", "linear")
POOL_NAMES = ("adaptive_avg", "adaptive_max", "identical", "none")


class Identical(nn.Module):
    def forward(self, input):
        return input


class Flatten(nn.Module):
    def forward(self, features):
        return features.reshape(features.shape[0], -1)


class Normalize(nn.Module):
    """L2 normalisation along ``dim`` (eps 1e-12, torch.nn.functional.normalize)"""

    def __init__(self, dim=1) -> None:
        super().__init__()
        self._dim = dim

    def forward(self, input):
        return nn.functional.normalize(input, p=2, dim=self._dim)


def _check_head_type(head_type) -> bool:
    return head_type in HEAD_TYPES


def _check_pool_name(pool_name) -> bool:
    return pool_name in POOL_NAMES


def get_pool_component(pool_name, spatial_size):
    if pool_name == "adaptive_avg":
        return nn.AdaptiveAvgPool2d(spatial_size)
    if pool_name == "adaptive_max":
        return nn.AdaptiveMaxPool2d(spatial_size)
    if pool_name in (None, "none", "identical"):
        return Identical()
    raise KeyError(pool_name)


class _ProjectorHeadBase(nn.Module):
    """argument checking and bookkeeping shared by the heads"""

    def __init__(self, *, input_dim: int, output_dim: int, head_type: str, normalize: bool, pool_name="adaptive_avg",
                 spatial_size=(1, 1)):
        super().__init__()
        assert _check_head_type(head_type=head_type)
        assert _check_pool_name(pool_name=pool_name)
        self._input_dim, self._output_dim = input_dim, output_dim
        self._head_type, self._normalize = head_type, normalize
        self._pool_name, self._spatial_size = pool_name, _pair(spatial_size)
        self._pooling_module = get_pool_component(self._pool_name, self._spatial_size)
