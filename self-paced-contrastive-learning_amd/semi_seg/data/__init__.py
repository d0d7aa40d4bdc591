"""On-device data path of the pre-train loop: slice store resident in HBM, the reference's
contrastive batch composition (``ContrastBatchSampler``), partition meta-labels, and the pre-train augmentation recipe
as one HIP launch per batch instead of PIL in DataLoader workers."""
from .rearr import ContrastBatchSampler, ContrastDataset  # noqa: F401
from .dataset import (ACDCSliceStore, DeviceSliceStore, ProstateSliceStore, acdc_partition, prostate_partition,  # noqa: F401
                      synthetic_slice_store)
from .augment import RECIPES, PretrainViews, RecipeViews, draw_view_params, resize_store  # noqa: F401
from .loader import (ContrastiveDeviceLoader, InfiniteRandomSampler, LabeledDeviceLoader,  # noqa: F401
                     get_contrastive_dataloader)
from .creator import (ScanBatchLoader, ScanBatchSampler, UnlabeledDeviceLoader, get_data, register_dataset,  # noqa: F401
                      set_data_root, split_dataset)
