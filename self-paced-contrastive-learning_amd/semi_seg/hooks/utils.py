"""Meta-label plumbing of the InfoNCE hooks, mirror of ``semi_seg/hooks/utils.py`` (``global_label_generator`` :9-42,
``get_label`` :45-65, ``meter_focus`` :68-74): which label generator a (dataset, ``contrast_on``) pair selects, how a
batch's file-name strings are split into the generator's arguments, and the decorator that scopes a hook method to
the hook's own meter group."""
import functools

from ..epochers.helper import ACDCCycleGenerator, PartitionLabelGenerator, PatientLabelGenerator, SIMCLRGenerator

# contrast_on -> generator class; "cycle" (cardiac phase) only exists for ACDC
_COMMON = {"partition": PartitionLabelGenerator, "patient": PatientLabelGenerator, "self": SIMCLRGenerator}
_GENERATORS = {"acdc": dict(_COMMON, cycle=ACDCCycleGenerator), "prostate": _COMMON, "prostate_md": _COMMON,
               "mmwhs": _COMMON}

# data_name -> (generator family, patient-id extractor, experiment-id extractor or None)
_NAME_RULES = {
    "acdc": ("acdc", lambda s: s.split("_")[0], lambda s: s.split("_")[1]),
    "prostate": ("prostate", lambda s: s.split("_")[0], None),
    "prostate_md": ("prostate", lambda s: s.split("_")[0], None),
    "mmwhsct": ("mmwhs", lambda s: s, None),
    "mmwhsmr": ("mmwhs", lambda s: s, None),
}


@functools.lru_cache()
def global_label_generator(dataset_name: str, contrast_on: str):
    try:
        family = _GENERATORS[dataset_name]
    except KeyError:
        raise NotImplementedError(dataset_name) from None
    if contrast_on not in family:
        raise NotImplementedError(contrast_on)
    return family[contrast_on]()


def get_label(contrast_on, data_name, partition_group, label_group):
    """integer meta-labels of a batch: ``label_group`` holds strings such as ``"patient004_00"`` (scan, phase)"""
    if data_name not in _NAME_RULES:
        raise NotImplementedError()
    family, patient_of, experiment_of = _NAME_RULES[data_name]
    arguments = {"partition_list": partition_group, "patient_list": [patient_of(s) for s in label_group]}
    if experiment_of is not None:
        arguments["experiment_list"] = [experiment_of(s) for s in label_group]
    return global_label_generator(dataset_name=family, contrast_on=contrast_on)(**arguments)


def meter_focus(method):
    """run a hook method with the epocher's meters focused on the hook's own group (``self._name``)"""

    @functools.wraps(method)
    def focused(self, *args, **kwargs):
        with self.meters.focus_on(self._name):
            return method(self, *args, **kwargs)

    return focused
