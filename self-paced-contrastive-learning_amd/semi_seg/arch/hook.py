"""Mirror of ``semi_seg/arch/hook.py``: forward-hook feature taps on UNet submodules (``_FeatureCollector`` :13-40,
``SingleFeatureExtractor`` :43-94, ``FeatureExtractor`` :97-143).  Pure host plumbing; the tapped tensors are the
HIP blocks' outputs (logical NCHW, channels-last storage)."""
from collections import OrderedDict
from contextlib import ExitStack, contextmanager
from typing import Iterator, List, Union

import torch

__all__ = ["FeatureExtractor", "SingleFeatureExtractor"]


class _FeatureCollector:
    def __init__(self, max_limit=5) -> None:
        self._count = 0
        self.feature = OrderedDict()
        self._enable = False
        self._max = max_limit

    def __call__(self, _, input_, result):
        if self._enable:
            self.feature[self._count] = result
            self._count += 1
            if self._count >= self._max:
                raise RuntimeError(f"You may forget to call clear as this hook "
                                   f"has registered data from {self._count} forward passes.")

    def clear(self):
        self._count = 0
        self.feature = OrderedDict()

    def set_enable(self, enable=True):
        self._enable = enable

    @property
    def enable(self):
        return self._enable


class SingleFeatureExtractor:
    def __init__(self, model, feature_name: str) -> None:
        self._model = model
        self._feature_name = feature_name
        assert self._feature_name in model.arch_elements, self._feature_name
        self._feature_extractor: _FeatureCollector = None
        self._hook_handler = None
        self._bound = False

    def bind(self):
        collector = _FeatureCollector()
        self._hook_handler = getattr(self._model, "_" + self._feature_name).register_forward_hook(collector)
        self._feature_extractor = collector
        self._bound = True

    def remove(self):
        self._hook_handler.remove()
        self._bound = False

    def __enter__(self):
        self.bind()
        return self

    def __exit__(self, *args, **kwargs):
        self.remove()

    def clear(self):
        self._feature_extractor.clear()

    def feature(self):
        feats = self._feature_extractor.feature
        if len(feats) > 0:
            vals = list(feats.values())
            return vals[0] if len(vals) == 1 else torch.cat(vals, dim=0)
        raise RuntimeError("no feature has been recorded.")

    def set_enable(self, enable=True):
        self._feature_extractor.set_enable(enable=enable)

    @contextmanager
    def enable_register(self, enable=True):
        prev = self._feature_extractor.enable
        self.set_enable(enable)
        yield
        self.set_enable(prev)


class FeatureExtractor:
    def __init__(self, model, feature_names: Union[str, List[str]]):
        self._feature_names = (feature_names,) if isinstance(feature_names, str) else feature_names
        self._extractor_list = [SingleFeatureExtractor(model, f) for f in self._feature_names]

    def bind(self):
        for e in self._extractor_list:
            e.bind()

    def remove(self):
        for e in self._extractor_list:
            e.remove()

    def __enter__(self):
        self.bind()
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.remove()

    def set_enable(self, enable=True):
        for e in self._extractor_list:
            e.set_enable(enable)

    @contextmanager
    def enable_register(self, enable=True):
        with ExitStack() as stack:
            for e in self._extractor_list:
                stack.enter_context(e.enable_register(enable=enable))
            yield

    def clear(self):
        for e in self._extractor_list:
            e.clear()

    def __iter__(self):
        for e in self._extractor_list:
            yield e.feature()

    def features(self) -> Iterator:
        return iter(self)

    def named_features(self) -> Iterator:
        for name, feature in zip(self._feature_names, self.features()):
            yield name, feature
