"""Mirror of fl_tissue_model_tools.transforms.filter_branch_seg_mask (reference transforms.py:306-361)."""
from __future__ import annotations

import numpy as np

from . import _lib

DISK2 = np.array([[0, 0, 1, 0, 0], [0, 1, 1, 1, 0], [1, 1, 1, 1, 1], [0, 1, 1, 1, 0], [0, 0, 1, 0, 0]], np.uint8)


def disk(radius: int):
    if radius != 2:
        raise NotImplementedError("only disk(2), the reference's footprint, is implemented")
    return DISK2.copy()


_handle = None


def default_handle() -> "_lib.Handle":
    """a model-less handle on device 0 (tmat_create_plain), created on first use"""
    global _handle
    if _handle is None:
        _handle = _lib.Handle(None, 0)
    return _handle


def filter_branch_seg_mask(mask, footprint=DISK2, remove_isolated=True, handle=None):
    """Remove components from the segmentation mask that do not contain branches.  `footprint`:
    disk(2) (default) or None to skip the median filter.  Returns a new bool mask.  Runs on the GPU
    (csrc/morph_kernels.hip through tmat_filter_mask_batch); a 3-D array filters a batch of masks."""
    if footprint is not None and not np.array_equal(np.asarray(footprint) != 0, DISK2 != 0):
        raise NotImplementedError("only footprint=disk(2) or None is implemented")
    m = np.asarray(mask) != 0
    h = handle or default_handle()
    if m.ndim == 3:
        return h.filter_mask(m, footprint is not None, remove_isolated)
    if m.ndim != 2:
        raise ValueError(f"expected a 2-D mask, got shape {m.shape}")
    return h.filter_mask(m[None], footprint is not None, remove_isolated)[0]
