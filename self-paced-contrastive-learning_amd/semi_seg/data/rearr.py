"""Mirror of ``semi_seg/data/rearr.py``: the batch composition of the contrastive pre-train loader.

``ContrastBatchSampler`` (:37-98): every batch draws ``scan_sample_num`` scans without replacement, then for each drawn
scan and each partition (in first-seen order) ``partition_sample_num`` slices of that scan in that partition; a
(scan, partition) pair with too few slices is skipped.  The draws consume python's ``random`` exactly as the reference
does (one ``random.sample`` over the scans, one per non-empty (scan, partition) pair over the SORTED slice indices), so
the same ``random.seed`` gives the same batches; the candidate lists are built once instead of intersecting two sets per
pair and batch."""
import random
from abc import ABCMeta, abstractmethod
from collections import OrderedDict
from typing import Callable, Dict, List, Union

__all__ = ["ContrastDataset", "ContrastBatchSampler"]


class ContrastDataset(metaclass=ABCMeta):
    """A slice knows its scan ("group") and its partition (position code along the scan); all scans share the
    partition alphabet (``rearr.py:12-34``)."""
    get_memory_dictionary: Callable[[], Dict[str, List[str]]]

    @abstractmethod
    def _get_partition(self, *args) -> Union[str, int]:
        ...

    @abstractmethod
    def show_partitions(self) -> List[Union[str, int]]:
        ...

    @abstractmethod
    def show_scan_names(self) -> List[Union[str, int]]:
        ...


class ContrastBatchSampler:
    class _SamplerIterator:
        def __init__(self, candidates, scans, scan_sample_num, partition_sample_num, shuffle):
            assert 1 <= scan_sample_num <= len(scans), scan_sample_num
            self._candidates, self._scans = candidates, scans
            self._scan_sample_num, self._partition_sample_num, self._shuffle = scan_sample_num, partition_sample_num, shuffle

        def __iter__(self):
            return self

        def __next__(self):
            batch = []
            for scan in random.sample(self._scans, self._scan_sample_num):
                for slices in self._candidates[scan]:
                    if len(slices) >= self._partition_sample_num:  # random.sample raises before drawing otherwise
                        batch.extend(random.sample(slices, self._partition_sample_num))
            if self._shuffle:
                random.shuffle(batch)
            return batch

    def __init__(self, dataset: ContrastDataset, scan_sample_num=4, partition_sample_num=1, shuffle=False) -> None:
        self._dataset = dataset
        filenames = list(next(iter(dataset.get_memory_dictionary().values())))
        scan2index, partition2index = OrderedDict(), OrderedDict()
        for i, filename in enumerate(filenames):
            scan2index.setdefault(dataset._get_scan_name(filename), []).append(i)  # noqa
            partition2index.setdefault(dataset._get_partition(filename), []).append(i)  # noqa
        self._scan2index, self._partition2index = scan2index, partition2index
        self._scans = list(scan2index.keys())
        # per scan, per partition (first-seen order): the sorted slice indices of that scan in that partition
        self._candidates = {}
        for scan, idx in scan2index.items():
            own = set(idx)
            self._candidates[scan] = [sorted(own.intersection(p)) for p in partition2index.values()]
        self._scan_sample_num, self._partition_sample_num, self._shuffle = scan_sample_num, partition_sample_num, shuffle

    def __iter__(self):
        return self._SamplerIterator(self._candidates, self._scans, self._scan_sample_num, self._partition_sample_num,
                                     self._shuffle)

    def __len__(self) -> int:
        return len(self._dataset)  # type: ignore
