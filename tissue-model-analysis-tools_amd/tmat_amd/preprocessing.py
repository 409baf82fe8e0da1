"""Mirror of the cell-area part of fl_tissue_model_tools.preprocessing / scripts/compute_cell_area.py (reference
preprocessing.py:44-93, compute_cell_area.py:29-87, 164-178) on the HIP library (csrc/cellarea_kernels.hip, no CPU fallback).

The reference thresholds one float image at a time with scikit-learn's GaussianMixture; here a batch of uint16 images goes
through tmat_cell_area_batch: bilinear down-sampling, rescale to 0..1, the mixture fitted to the intensity histogram,
threshold, area.  The fit is deterministic (optimal 2-means start instead of sklearn's RandomState-seeded KMeans; float64
EM), so it does not need the reference's `rand_state`; tests/test_oracle_cellarea.py states the tolerance against
scikit-learn (0.1 percentage points of the image area).
"""
from __future__ import annotations

import numpy as np

from ._lib import Handle, check, lib, ptr


def resized_shape(shape, dsamp_size):
    """compute_cell_area.py:54-57: dsize = round(shape * dsamp_size / max(shape)) is handed to cv2 as (width, height)"""
    ratio = dsamp_size / max(shape)
    dsize = tuple(int(v) for v in np.round(np.multiply(shape, ratio)).astype(int))
    return dsize[1], dsize[0]


class _source_depth:
    """cv2.resize runs uint8 images through its fixed-point bilinear arithmetic and uint16 images through its float arithmetic
    (oracle/cellarea.py:resize_linear_u8): the entry points take widened uint16 arrays, the handle carries the source depth"""

    def __init__(self, handle: Handle, dtype):
        self.handle, self.bits = handle, 8 if np.dtype(dtype) == np.uint8 else 16

    def __enter__(self):
        check(lib().tmat_set_input_depth(self.handle.raw, self.bits), "tmat_set_input_depth")

    def __exit__(self, *exc):
        check(lib().tmat_set_input_depth(self.handle.raw, 16), "tmat_set_input_depth")
        return False


def cell_area_batch(handle: Handle, imgs: np.ndarray, dsamp_size=512, sd_coef: float = 0.0, return_params=False):
    """imgs (n, H, W) uint8 / uint16 -> (area fractions (n,), thresholded images (n, h, w) uint8 0 / 255[, fit parameters (n, 9)])"""
    a = np.asarray(imgs)
    if a.ndim != 3 or a.dtype not in (np.uint8, np.uint16):
        raise ValueError("cell_area_batch: expected (n, H, W) uint8 or uint16 images")
    src_dtype = a.dtype
    a = np.ascontiguousarray(a, np.uint16)
    n, H, W = a.shape
    oh, ow = resized_shape((H, W), dsamp_size) if dsamp_size is not None else (0, 0)
    area = np.empty(n, np.float64)
    out = np.empty((n, oh or H, ow or W), np.uint8)
    params = np.empty((n, 9), np.float64)
    with _source_depth(handle, src_dtype):
        check(lib().tmat_cell_area_batch(handle.raw, ptr(a), n, H, W, oh, ow, float(sd_coef), ptr(area), ptr(out), ptr(params)), "tmat_cell_area_batch")
    return (area, out, params) if return_params else (area, out)


def resize_batch(handle: Handle, imgs: np.ndarray, dsamp_size) -> np.ndarray:
    """the down-sampling step of load_img (compute_cell_area.py:54-57) for a batch, on the device: (n, H, W) -> (n, h, w) uint16
    (uint8 sources: cv2's fixed-point arithmetic, values stay <= 255)"""
    src_dtype = np.asarray(imgs).dtype
    a = np.ascontiguousarray(imgs, np.uint16)
    n, H, W = a.shape
    oh, ow = resized_shape((H, W), dsamp_size)
    out = np.empty((n, oh, ow), np.uint16)
    with _source_depth(handle, src_dtype):
        check(lib().tmat_resize_linear_u16(handle.raw, ptr(a), n, H, W, oh, ow, ptr(out)), "tmat_resize_linear_u16")
    return out


def cell_area_batch_well(handle: Handle, imgs: np.ndarray, dsamp_size=512, sd_coef: float = 0.0, well_seed: int = 0):
    """--detect-well form (compute_cell_area.py:117-130, 273-286): down-sample, well mask of every down-sampled image
    (generate_well_mask(img, mask_val=255)), mixture fitted inside the mask, area relative to the well's pixel count.
    -> (area fractions (n,), thresholded (n, h, w) uint8 0 / 255, well masks (n, h, w) uint8 0 / 255)"""
    from . import well_mask_generation as wmg
    a = np.asarray(imgs)
    if a.ndim != 3 or a.dtype not in (np.uint8, np.uint16):
        raise ValueError("cell_area_batch_well: expected (n, H, W) uint8 or uint16 images")
    src_dtype = a.dtype
    small = resize_batch(handle, a, dsamp_size) if dsamp_size is not None else np.ascontiguousarray(a, np.uint16)
    n, h, w = small.shape
    # the reference makes the mask from the image in its own dtype (uint8 images stay uint8 through cv2.resize)
    masks = np.stack([np.asarray(wmg.generate_well_mask(small[i].astype(src_dtype), mask_val=255, handle=handle, seed=well_seed)) for i in range(n)])
    m8 = np.ascontiguousarray(masks > 0, np.uint8)
    area = np.empty(n, np.float64)
    out = np.empty((n, h, w), np.uint8)
    params = np.empty((n, 9), np.float64)
    check(lib().tmat_cell_area_masked(handle.raw, ptr(small), ptr(m8), n, h, w, float(sd_coef), ptr(area), ptr(out), ptr(params)), "tmat_cell_area_masked")
    # compute_area_prop(img, well_pix_area): np.sum(img > 0) / ref_area, from the integer counts
    kept_px = (out > 0).reshape(n, -1).sum(1)
    well_px = m8.reshape(n, -1).sum(1)
    return kept_px / well_px, out, masks.astype(np.uint8)


def exec_threshold(handle: Handle, img: np.ndarray, sd_coef: float = 0.0) -> np.ndarray:
    """preprocessing.exec_threshold for one integer image without a well mask: the image with background pixels set to 0"""
    area, kept = cell_area_batch(handle, np.asarray(img)[None], None, sd_coef)
    return np.where(kept[0] > 0, img, 0)
