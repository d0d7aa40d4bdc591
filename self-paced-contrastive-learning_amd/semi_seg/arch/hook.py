"""Feature taps on UNet submodules.

Mirror of the reference's ``semi_seg/arch/hook.py`` -- ``SingleFeatureExtractor`` (:43-94) and ``FeatureExtractor``
(:97-143) keep their names and methods (``bind`` / ``remove`` / ``clear`` / ``set_enable`` / ``enable_register`` /
``feature`` / ``features`` / ``named_features``, context-manager use) -- implemented on one small recorder object.

A tap is a PyTorch forward hook on the submodule ``_<name>``.  While recording is switched on it keeps the module's
output of every forward pass; keeping ``max_limit`` passes without a ``clear()`` is treated as a forgotten ``clear()``
and raises, like the reference's ``_FeatureCollector`` (:13-40).  Host plumbing only: what is recorded are the HIP
blocks' output tensors (logical NCHW over channels-last storage), which never leave the device.
"""
import contextlib

import torch

__all__ = ["FeatureExtractor", "SingleFeatureExtractor"]


class _FeatureCollector:
    def __init__(self, max_limit: int = 5) -> None:
        self.feature = {}          # index of the forward pass -> recorded output, in order
        self._recording = False
        self._max_passes = max_limit

    # forward-hook signature of torch.nn.Module.register_forward_hook
    def __call__(self, module, inputs, output):
        if self._recording:
            self.feature[len(self.feature)] = output
            if len(self.feature) >= self._max_passes:
                raise RuntimeError("You may forget to call clear as this hook has registered data from "
                                   f"{len(self.feature)} forward passes.")

    @property
    def enable(self):
        return self._recording

    def set_enable(self, enable=True):
        self._recording = bool(enable)

    def clear(self):
        self.feature = {}


class SingleFeatureExtractor:
    def __init__(self, model, feature_name: str) -> None:
        if feature_name not in model.arch_elements:
            raise AssertionError(feature_name)
        self._model = model
        self._feature_name = feature_name
        self._feature_extractor = None  # the _FeatureCollector of the current binding
        self._hook_handler = None       # torch's RemovableHandle

    # ---- binding
    def bind(self):
        collector = _FeatureCollector()
        self._hook_handler = getattr(self._model, f"_{self._feature_name}").register_forward_hook(collector)
        self._feature_extractor = collector

    def remove(self):
        self._hook_handler.remove()

    def __enter__(self):
        self.bind()
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self.remove()

    # ---- recording
    def set_enable(self, enable=True):
        self._feature_extractor.set_enable(enable)

    def clear(self):
        self._feature_extractor.clear()

    @contextlib.contextmanager
    def enable_register(self, enable=True):
        previous = self._feature_extractor.enable
        self._feature_extractor.set_enable(enable)
        try:
            yield
        finally:
            self._feature_extractor.set_enable(previous)

    def feature(self):
        passes = tuple(self._feature_extractor.feature.values())
        if len(passes) == 0:
            raise RuntimeError("no feature has been recorded.")
        if len(passes) == 1:
            return passes[0]  # the usual case: no concatenation copy
        return torch.cat(passes, dim=0)


class FeatureExtractor:
    """taps on several submodules, driven together"""

    def __init__(self, model, feature_names):
        names = (feature_names,) if isinstance(feature_names, str) else feature_names
        self._feature_names = names
        self._extractor_list = [SingleFeatureExtractor(model, n) for n in names]

    def __iter__(self):
        for tap in self._extractor_list:
            yield tap.feature()

    def features(self):
        return iter(self)

    def named_features(self):
        return zip(self._feature_names, self)

    def bind(self):
        [tap.bind() for tap in self._extractor_list]

    def remove(self):
        [tap.remove() for tap in self._extractor_list]

    def clear(self):
        [tap.clear() for tap in self._extractor_list]

    def set_enable(self, enable=True):
        [tap.set_enable(enable) for tap in self._extractor_list]

    def __enter__(self):
        self.bind()
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self.remove()

    @contextlib.contextmanager
    def enable_register(self, enable=True):
        with contextlib.ExitStack() as stack:
            [stack.enter_context(tap.enable_register(enable=enable)) for tap in self._extractor_list]
            yield
