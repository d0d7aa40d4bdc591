from .unet import UNet, arch_order, get_channel_dim, sort_arch  # noqa: F401
from .hook import FeatureExtractor, SingleFeatureExtractor  # noqa: F401
