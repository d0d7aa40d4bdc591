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

# reference import name -> mirror module (relative to this package).  Packages first: ``from semi_seg.hooks import x``
# resolves ``semi_seg`` and ``semi_seg.hooks`` through sys.modules before it looks at the file system.
_MIRRORED = {
    "contrastyou": "contrastyou",
    "contrastyou.losses": "contrastyou.losses",
    "contrastyou.losses.contrast_loss3": "contrastyou.losses.contrast_loss3",
    "contrastyou.meters": "contrastyou.meters",
    "contrastyou.projectors": "contrastyou.projectors",
    "contrastyou.projectors.heads": "contrastyou.projectors.heads",
    "contrastyou.projectors.nn": "contrastyou.projectors.nn",
    "contrastyou.hooks": "contrastyou.hooks",
    "contrastyou.hooks.base": "contrastyou.hooks.base",
    "semi_seg": "semi_seg",
    "semi_seg.arch": "semi_seg.arch",
    "semi_seg.arch.unet": "semi_seg.arch.unet",
    "semi_seg.arch.hook": "semi_seg.arch.hook",
    "semi_seg.hooks": "semi_seg.hooks",
    "semi_seg.hooks.creator": "semi_seg.hooks.creator",
    "semi_seg.hooks.infonce": "semi_seg.hooks.infonce",
    "semi_seg.hooks.utils": "semi_seg.hooks.utils",
    "semi_seg.epochers": "semi_seg.epochers",
    "semi_seg.epochers.new_pretrain": "semi_seg.epochers.pretrain",
    "semi_seg.epochers.new_epocher": "semi_seg.epochers.finetune",
    "semi_seg.epochers.helper": "semi_seg.epochers.helper",
    "semi_seg.trainers": "semi_seg.trainers",
    "semi_seg.trainers.new_pretrain": "semi_seg.trainers.pretrain",
    "semi_seg.trainers.new_trainer": "semi_seg.trainers.finetune",
    "semi_seg.data": "semi_seg.data",
    "semi_seg.data.rearr": "semi_seg.data.rearr",
    "semi_seg.data.creator": "semi_seg.data.creator",
    "val": "val",
    "hook_creator": "hook_creator",
}


def install(strict: bool = False, reference_root: str = None):
    """Register the mirror modules under the reference's import names (``semi_seg.arch``, ``semi_seg.hooks``,
    ``semi_seg.trainers.new_pretrain``, ``contrastyou.losses.contrast_loss3``, ``hook_creator`` ...) so that the
    reference's drivers -- the import lines and the body of ``main_pretrain_encoder.worker`` -- pick up the HIP-backed
    classes unchanged.  Call before importing them.

    ``reference_root``: a checkout of the reference.  Its ``semi_seg`` / ``contrastyou`` directories are appended to the
    mirror packages' search paths, so every module that is NOT mirrored (data sets, configuration, writers) still comes
    from the reference while the hot path comes from here.  ``deepclustering2.loss.KL_div`` (un-vendored third party) is
    provided by the mirror's restatement only when that package cannot be imported."""
    import importlib
    import os
    import sys
    import types
    done = []
    for name, rel in _MIRRORED.items():
        try:
            mod = importlib.import_module(f"{__name__}.{rel}")
        except ImportError:
            if strict:
                raise
            continue
        sys.modules[name] = mod
        done.append(name)
    if reference_root:
        for top in ("semi_seg", "contrastyou"):
            d = os.path.join(reference_root, top)
            pkg = sys.modules.get(top)
            if pkg is not None and os.path.isdir(d) and d not in pkg.__path__:
                pkg.__path__.append(d)
        if reference_root not in sys.path:
            sys.path.append(reference_root)
    try:
        importlib.import_module("deepclustering2.loss")
    except Exception:  # noqa: BLE001  (absent, or present but broken on this Python / torch)
        from .contrastyou.losses import kl as _kl
        top = sys.modules.setdefault("deepclustering2", types.ModuleType("deepclustering2"))
        loss = types.ModuleType("deepclustering2.loss")
        loss.KL_div = _kl.KL_div
        top.loss = loss
        sys.modules["deepclustering2.loss"] = loss
        done.append("deepclustering2.loss")
    return done
