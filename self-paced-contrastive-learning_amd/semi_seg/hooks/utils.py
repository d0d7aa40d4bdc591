"""Mirror of ``semi_seg/hooks/utils.py``: ``get_label`` (:45-65), ``global_label_generator`` (:9-42),
``meter_focus`` (:68-74)."""
from functools import lru_cache, wraps

from ..epochers.helper import PartitionLabelGenerator, PatientLabelGenerator, ACDCCycleGenerator, SIMCLRGenerator


@lru_cache()
def global_label_generator(dataset_name: str, contrast_on: str):
    table = {"partition": PartitionLabelGenerator, "patient": PatientLabelGenerator, "self": SIMCLRGenerator}
    if dataset_name == "acdc":
        table = dict(table, cycle=ACDCCycleGenerator)
    elif dataset_name not in ("prostate", "prostate_md", "mmwhs"):
        raise NotImplementedError(dataset_name)
    if contrast_on not in table:
        raise NotImplementedError(contrast_on)
    return table[contrast_on]()


def get_label(contrast_on, data_name, partition_group, label_group):
    if data_name == "acdc":
        return global_label_generator(dataset_name="acdc", contrast_on=contrast_on)(
            partition_list=partition_group, patient_list=[p.split("_")[0] for p in label_group],
            experiment_list=[p.split("_")[1] for p in label_group])
    elif data_name in ("prostate", "prostate_md"):
        return global_label_generator(dataset_name="prostate", contrast_on=contrast_on)(
            partition_list=partition_group, patient_list=[p.split("_")[0] for p in label_group])
    elif data_name in ("mmwhsct", "mmwhsmr"):
        return global_label_generator(dataset_name="mmwhs", contrast_on=contrast_on)(
            partition_list=partition_group, patient_list=label_group)
    raise NotImplementedError()


def meter_focus(func):
    @wraps(func)
    def func_wrapper(self, *args, **kwargs):
        with self.meters.focus_on(self._name):
            return func(self, *args, **kwargs)

    return func_wrapper
