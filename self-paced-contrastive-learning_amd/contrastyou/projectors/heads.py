"""HIP-backed mirror of ``contrastyou/projectors/heads.py:9-25`` (get_contrastive_projector) and ``:78-92``
(ProjectionHead): global average pool -> Linear -> LeakyReLU(0.01) -> Linear -> L2 normalise, one fused call
(csrc/projector.hip).  Parameters live in ``_header`` at the reference's Sequential indices so checkpoints
interchange (``_header.2.{weight,bias}``, ``_header.4.{weight,bias}``)."""
from torch import nn

from ... import functional as F_hip
from .nn import _ProjectorHeadBase, Flatten, Normalize, Identical

__all__ = ["ProjectionHead", "get_contrastive_projector"]


def get_contrastive_projector(*, head_type: str, pool_module, input_dim, hidden_dim, output_dim, normalize: bool):
    if head_type == "mlp":
        return nn.Sequential(pool_module, Flatten(), nn.Linear(input_dim, hidden_dim),
                             nn.LeakyReLU(0.01, inplace=True), nn.Linear(hidden_dim, output_dim),
                             Normalize() if normalize else Identical())
    return nn.Sequential(pool_module, Flatten(), nn.Linear(input_dim, output_dim),
                         Normalize() if normalize else Identical())


class ProjectionHead(_ProjectorHeadBase):
    def __init__(self, *, input_dim: int, hidden_dim=256, output_dim: int, head_type: str, normalize: bool,
                 pool_name="adaptive_avg", spatial_size=(1, 1)):
        assert pool_name in ("adaptive_avg", "adaptive_max")
        super().__init__(input_dim=input_dim, output_dim=output_dim, head_type=head_type, normalize=normalize,
                         pool_name=pool_name, spatial_size=spatial_size)
        if pool_name != "adaptive_avg" or tuple(self._spatial_size) != (1, 1):
            raise NotImplementedError("the HIP projector implements the hot-path configuration only: adaptive_avg "
                                      f"pooling to (1,1) (semi_seg/hooks/infonce.py:96-99); got {pool_name} "
                                      f"{self._spatial_size}")
        self._header = get_contrastive_projector(head_type=self._head_type, pool_module=self._pooling_module,
                                                 input_dim=self._input_dim, hidden_dim=hidden_dim,
                                                 output_dim=output_dim, normalize=normalize)

    def forward(self, features):
        h = self._header
        if self._head_type == "mlp":
            return F_hip.projector(features, h[2].weight, h[2].bias, h[4].weight, h[4].bias, self._normalize)
        return F_hip.projector(features, h[2].weight, h[2].bias, None, None, self._normalize)
