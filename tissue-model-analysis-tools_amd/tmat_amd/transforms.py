"""Mirror of fl_tissue_model_tools.transforms.filter_branch_seg_mask (reference transforms.py:306-361)."""
from __future__ import annotations

import numpy as np

from . import _lib

DISK2 = np.array([[0, 0, 1, 0, 0], [0, 1, 1, 1, 0], [1, 1, 1, 1, 1], [0, 1, 1, 1, 0], [0, 0, 1, 0, 0]], np.uint8)


def disk(radius: int):
    if radius != 2:
        raise NotImplementedError("only disk(2), the reference's footprint, is implemented")
    return DISK2.copy()


def filter_branch_seg_mask(mask, footprint=DISK2, remove_isolated=True):
    """Remove components from the segmentation mask that do not contain branches.  `footprint`:
    disk(2) (default) or None to skip the median filter.  Returns a new bool mask."""
    if footprint is not None and not np.array_equal(np.asarray(footprint) != 0, DISK2 != 0):
        raise NotImplementedError("only footprint=disk(2) or None is implemented")
    return _lib.host_filter_mask(np.asarray(mask) != 0, footprint is not None, remove_isolated)
