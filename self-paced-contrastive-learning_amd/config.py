"""Process-wide defaults of the HIP path."""
import os

import torch

_DTYPES = {"fp32": torch.float32, "f32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16,
           "bfloat16": torch.bfloat16}
_compute_dtype = _DTYPES[os.environ.get("SPCL_COMPUTE_DTYPE", "fp32").lower()]


def set_compute_dtype(dtype):
    """Activation/weight storage dtype of newly built (and, via UNet.set_compute_dtype, existing) encoders.
    torch.float32 = parity mode (exact-f32 MFMA); torch.bfloat16 = bf16 storage, f32 accumulation/statistics."""
    global _compute_dtype
    if isinstance(dtype, str):
        dtype = _DTYPES[dtype.lower()]
    assert dtype in (torch.float32, torch.bfloat16), dtype
    _compute_dtype = dtype


def get_compute_dtype():
    return _compute_dtype
