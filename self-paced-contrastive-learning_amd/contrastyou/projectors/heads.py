"""HIP-backed mirror of ``contrastyou/projectors/heads.py``: ``get_contrastive_projector`` (:9-25) / ``ProjectionHead``
(:78-92) -- global pool -> Linear -> LeakyReLU(0.01) -> Linear -> L2 normalise, one fused call (csrc/projector.hip) -- and
``get_contrastive_dense_projector`` (:28-39) / ``DenseProjectionHead`` (:96-120) -- 1x1-conv MLP on every pixel ->
adaptive pool to ``spatial_size`` -> L2 normalise over channels.  Parameters live at the reference's ``nn.Sequential``
indices so checkpoints interchange (``_header.2.*`` / ``_header.4.*``; ``_projector.0.*`` / ``_projector.2.*``)."""
from torch import nn

from ... import functional as F_hip
from .nn import _ProjectorHeadBase, Flatten, Normalize, Identical

__all__ = ["ProjectionHead", "DenseProjectionHead", "get_contrastive_projector", "get_contrastive_dense_projector"]


def get_contrastive_projector(*, head_type: str, pool_module, input_dim, hidden_dim, output_dim, normalize: bool):
    if head_type == "mlp":
        return nn.Sequential(pool_module, Flatten(), nn.Linear(input_dim, hidden_dim),
                             nn.LeakyReLU(0.01, inplace=True), nn.Linear(hidden_dim, output_dim),
                             Normalize() if normalize else Identical())
    return nn.Sequential(pool_module, Flatten(), nn.Linear(input_dim, output_dim),
                         Normalize() if normalize else Identical())


def get_contrastive_dense_projector(*, head_type: str, input_dim, hidden_dim, output_dim):
    if head_type == "mlp":
        return nn.Sequential(nn.Conv2d(input_dim, hidden_dim, 1, 1, 0), nn.LeakyReLU(0.01, inplace=True),
                             nn.Conv2d(hidden_dim, output_dim, 1, 1, 0))
    return nn.Sequential(nn.Conv2d(input_dim, output_dim, 1, 1, 0))


class ProjectionHead(_ProjectorHeadBase):
    """``pool_name="adaptive_max"`` pools with the HIP adaptive-max kernel first and hands the [N, C] rows to the fused
    projector.  A ``spatial_size`` other than (1, 1) is accepted by the constructor like the reference's, and fails in
    ``forward`` like the reference's: ``Flatten`` would hand ``C * h * w`` features to ``Linear(C, ...)``."""

    def __init__(self, *, input_dim: int, hidden_dim=256, output_dim: int, head_type: str, normalize: bool,
                 pool_name="adaptive_avg", spatial_size=(1, 1)):
        assert pool_name in ("adaptive_avg", "adaptive_max")
        super().__init__(input_dim=input_dim, output_dim=output_dim, head_type=head_type, normalize=normalize,
                         pool_name=pool_name, spatial_size=spatial_size)
        self._header = get_contrastive_projector(head_type=self._head_type, pool_module=self._pooling_module,
                                                 input_dim=self._input_dim, hidden_dim=hidden_dim,
                                                 output_dim=output_dim, normalize=normalize)

    def forward(self, features, normalize=None):
        """``normalize`` (not in the reference): overrides the constructor's setting for this call -- ``False`` returns the
        rows before ``F.normalize`` for a criterion that normalises in its own launch (``normalize_inputs=True``)"""
        norm = self._normalize if normalize is None else bool(normalize)
        h = self._header
        if tuple(self._spatial_size) != (1, 1):
            k = self._spatial_size[0] * self._spatial_size[1]
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({features.shape[0]}x{self._input_dim * k} and "
                               f"{self._input_dim}x{h[2].out_features})")
        if self._pool_name == "adaptive_max":
            features = F_hip.adaptive_pool2d(features, (1, 1), "max")
        if self._head_type == "mlp":
            return F_hip.projector(features, h[2].weight, h[2].bias, h[4].weight, h[4].bias, norm)
        return F_hip.projector(features, h[2].weight, h[2].bias, None, None, norm)


class DenseProjectionHead(_ProjectorHeadBase):
    """heads.py:96-120: the pixel-wise projection of a decoder feature map"""

    def __init__(self, *, input_dim: int, hidden_dim=128, output_dim: int, head_type: str, normalize: bool,
                 pool_name="adaptive_avg", spatial_size=(16, 16)):
        super().__init__(input_dim=input_dim, output_dim=output_dim, head_type=head_type, normalize=normalize,
                         pool_name=pool_name, spatial_size=spatial_size)
        self._projector = get_contrastive_dense_projector(head_type=self._head_type, input_dim=self._input_dim,
                                                          hidden_dim=hidden_dim, output_dim=output_dim)

    def forward(self, features):
        p = self._projector
        if (self._head_type == "mlp" and self._pool_name == "adaptive_avg"
                and F_hip.pixelwise_mlp_pooled_supported(features, p[0].weight, p[2].weight)):
            # average pooling commutes with the (linear) second layer: pool the hidden activation, project the pooled rows
            out = F_hip.pixelwise_mlp_pooled(features, p[0].weight, p[0].bias, p[2].weight, p[2].bias, self._spatial_size)
            return F_hip.l2norm_channels(out) if self._normalize else out
        if self._head_type == "mlp":
            out = F_hip.pixelwise_mlp(features, p[0].weight, p[0].bias, p[2].weight, p[2].bias)
        else:
            out = F_hip.pixelwise_mlp(features, p[0].weight, p[0].bias)
        if self._pool_name in ("adaptive_avg", "adaptive_max"):  # "change resolution here" (:111-112)
            out = F_hip.adaptive_pool2d(out, self._spatial_size, "max" if self._pool_name == "adaptive_max" else "avg")
        if self._normalize:
            return F_hip.l2norm_channels(out)
        return out
