"""Mirror of the InfoNCE factories in ``semi_seg/hooks/creator.py``: ``feature_until_from_hooks`` (:23-29),
``create_infonce_hooks`` (:69-99) and ``create_sp_infonce_hooks`` (:102-124) -- one hook per (feature, weight,
contrast_on) triple combined into one TrainerHook."""
from typing import List, Union

from ...contrastyou.hooks.base import CombineTrainerHook
from ..arch.unet import sort_arch
from .infonce import INFONCEHook, SelfPacedINFONCEHook


def _listify(v, n):
    return list(v) if isinstance(v, (list, tuple)) else [v] * n


def feature_until_from_hooks(*hooks) -> Union[str, None]:
    names = []
    for h in hooks:
        for sub in (h._hooks if isinstance(h, CombineTrainerHook) else [h]):
            if hasattr(sub, "_feature_name"):
                names.append(sub._feature_name)
    return sort_arch(names)[-1] if names else None


def create_infonce_hooks(*, model, feature_names: Union[str, List[str]], weights, contrast_ons, data_name="acdc",
                         **kw):
    feature_names = _listify(feature_names, 1)
    n = len(feature_names)
    hooks = [INFONCEHook(name=f"infonce_{f.lower()}_{c}", model=model, feature_name=f, weight=w, data_name=data_name,
                         contrast_on=c, **kw)
             for f, w, c in zip(feature_names, _listify(weights, n), _listify(contrast_ons, n))]
    return CombineTrainerHook(*hooks)


def create_sp_infonce_hooks(*, model, feature_names: Union[str, List[str]], weights, contrast_ons, begin_values,
                            end_values, mode: str, max_epoch: int, p=0.5, correct_grad=False, data_name="acdc", **kw):
    feature_names = _listify(feature_names, 1)
    n = len(feature_names)
    corr = _listify(correct_grad, n)
    hooks = [SelfPacedINFONCEHook(name=f"spinfonce_{f.lower()}_{c}", model=model, feature_name=f, weight=w,
                                  data_name=data_name, contrast_on=c, mode=mode, p=p, begin_value=b, end_value=e,
                                  correct_grad=cg, max_epoch=max_epoch, **kw)
             for f, w, c, b, e, cg in zip(feature_names, _listify(weights, n), _listify(contrast_ons, n),
                                          _listify(begin_values, n), _listify(end_values, n), corr)]
    return CombineTrainerHook(*hooks)
