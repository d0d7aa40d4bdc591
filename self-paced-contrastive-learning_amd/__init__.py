"""MI355X-native (gfx950) implementation of the self-paced contrastive pre-train hot path of
jizongFox/Self-paced-Contrastive-Learning, behind the reference's own class API.

    import spcl_amd                                   # via the root shim spcl_amd.py
    from spcl_amd.semi_seg.arch import UNet
    from spcl_amd.contrastyou.losses.contrast_loss3 import SelfPacedSupConLoss
    spcl_amd.install()                                # alias the mirror modules under the reference's names

Compute = hand-written HIP kernels in csrc/ behind the C ABI of include/spcl_hip.h (ctypes, native.py).
"""
from . import native  # noqa: F401

__version__ = "0.1.0"

_MIRRORED = (
    "contrastyou.losses.contrast_loss3",
    "contrastyou.meters",
    "contrastyou.projectors.heads",
    "contrastyou.projectors.nn",
    "contrastyou.hooks.base",
    "semi_seg.arch.unet",
    "semi_seg.arch.hook",
    "semi_seg.hooks.infonce",
    "semi_seg.hooks.utils",
)


def install(strict: bool = False):
    """Register the mirror modules under the reference's import names (``contrastyou.losses.contrast_loss3`` ...)
    so that the reference's drivers pick up the HIP-backed classes unchanged.  Call before importing them."""
    import importlib
    import sys
    done = []
    for name in _MIRRORED:
        try:
            mod = importlib.import_module(f"{__name__}.{name}")
        except ImportError:
            if strict:
                raise
            continue
        sys.modules[name] = mod
        done.append(name)
    return done
