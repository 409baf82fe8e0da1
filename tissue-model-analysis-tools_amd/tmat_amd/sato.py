"""Host side of the Z-stack (Sato) branch of analyze_img (reference scripts/compute_branches.py:224-306).

The reference does this branch inline in analyze_img with scikit-image / scipy calls; here the pixel work runs in
libtmat_hip.so (csrc/sato_kernels.hip, csrc/stack_pipeline.cpp) and this module is the thin host mirror: it hands scipy's
gaussian kernel tables to the library, calls the C-ABI and shapes the results.  No CPU fallback.

Gaussian tables: scipy builds them with numpy.exp, whose float64 loop is CPU-dispatched (its AVX512 path and libm's exp
differ in the last bit at some taps).  `install_gaussian_tables` computes the tables exactly as
scipy/ndimage/_filters.py:_gaussian_kernel1d does -- with THIS host's numpy -- and registers them on the handle, so the
filters reproduce what the reference would compute on the same machine bit for bit.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from ._lib import Handle, Row, VesselStages, check, lib, ptr

SATO_SIGMAS = (1, 2, 3, 4, 5, 7, 9, 11, 13, 15)            # compute_branches.py:262
GAUSSIAN_DERIVATIVES, GRADIENT = 0, 1                      # TMAT_SATO_*: scikit-image >= 0.20 (the pinned 0.22.0) / <= 0.19
HESSIAN_FORMS = {"gaussian_derivatives": GAUSSIAN_DERIVATIVES, "gradient": GRADIENT}
NEAREST, REFLECT, MIRROR = 0, 1, 2


def gaussian_kernel1d(sigma: float, order: int, radius: int) -> np.ndarray:
    """scipy.ndimage._filters._gaussian_kernel1d(sigma, order, radius)[::-1], the weights gaussian_filter1d hands to
    correlate1d (scipy 1.13: _filters.py:177-206, 265-268)"""
    if order < 0:
        raise ValueError("order must be non-negative")
    exponent_range = np.arange(order + 1)
    sigma2 = sigma * sigma
    x = np.arange(-radius, radius + 1)
    phi_x = np.exp(-0.5 / sigma2 * x ** 2)
    phi_x = phi_x / phi_x.sum()
    if order == 0:
        return np.ascontiguousarray(phi_x[::-1], np.float64)
    q = np.zeros(order + 1)
    q[0] = 1
    D = np.diag(exponent_range[1:], 1)
    P = np.diag(np.ones(order) / -sigma2, -1)
    Q_deriv = D + P
    for _ in range(order):
        q = Q_deriv.dot(q)
    q = (x[:, None] ** exponent_range).dot(q)
    return np.ascontiguousarray((q * phi_x)[::-1], np.float64)


def ensure_gaussian_table(handle: Handle, sigma: float, order: int, truncate: float):
    """register scipy's table for (sigma, order, radius = int(truncate sigma + 0.5)) on the handle, once"""
    done = getattr(handle, "_gauss_done", None)
    if done is None:
        done = handle._gauss_done = set()
    radius = int(truncate * float(sigma) + 0.5)
    key = (float(sigma), int(order), radius)
    if key in done:
        return
    w = gaussian_kernel1d(float(sigma), order, radius)
    check(lib().tmat_set_gaussian_table(handle.raw, float(sigma), order, radius, ptr(w)), "tmat_set_gaussian_table")
    done.add(key)


def install_gaussian_tables(handle: Handle, stack_hw=None, out_hw=None, sigmas=SATO_SIGMAS):
    """register every table the branch uses: sigma 1 and 2 (truncate 4), the Sato scales in both Hessian forms, and the
    anti-aliasing sigmas of the resize from stack_hw to out_hw"""
    ensure_gaussian_table(handle, 1.0, 0, 4.0)
    ensure_gaussian_table(handle, 2.0, 0, 4.0)
    sq1_2 = 1 / math.sqrt(2)
    for s in sigmas:
        ensure_gaussian_table(handle, float(s), 0, 4.0)          # gradient form
        tr = 8.0 if s > 1 else 100.0                             # corner.py:_hessian_matrix_with_gaussian
        ensure_gaussian_table(handle, sq1_2 * s, 0, tr)
        ensure_gaussian_table(handle, sq1_2 * s, 1, tr)
    if stack_hw is not None and out_hw is not None:
        for n_in, n_out in zip(stack_hw, out_hw):
            sg = max(0.0, (n_in / n_out - 1) / 2)
            if sg > 1e-15:
                ensure_gaussian_table(handle, sg, 0, 4.0)


def gaussian(handle: Handle, x: np.ndarray, sigma=1.0, mode=NEAREST) -> np.ndarray:
    """skimage.filters.gaussian on a float32 array of 2 or 3 dimensions"""
    x = np.ascontiguousarray(x, np.float32)
    if x.ndim not in (2, 3):
        raise ValueError("gaussian: 2-D or 3-D float32 input")
    ensure_gaussian_table(handle, sigma, 0, 4.0)
    d = (1,) + x.shape if x.ndim == 2 else x.shape
    out = np.empty_like(x)
    check(lib().tmat_gaussian_f32(handle.raw, ptr(x), d[0], d[1], d[2], float(sigma), int(mode), ptr(out)), "tmat_gaussian_f32")
    return out


def sato(handle: Handle, imgs: np.ndarray, sigmas=SATO_SIGMAS, hessian="gaussian_derivatives") -> np.ndarray:
    """skimage.filters.sato(img, sigmas, black_ridges=False) on one (h, w) or n (n, h, w) float32 images"""
    x = np.ascontiguousarray(imgs, np.float32)
    single = x.ndim == 2
    xb = x[None] if single else x
    install_gaussian_tables(handle, sigmas=tuple(sigmas))
    sg = np.ascontiguousarray(sigmas, np.float64)
    out = np.empty_like(xb)
    check(lib().tmat_sato_batch(handle.raw, ptr(xb), xb.shape[0], xb.shape[1], xb.shape[2], ptr(sg), len(sg), HESSIAN_FORMS[hessian], ptr(out)),
          "tmat_sato_batch")
    return out[0] if single else out


def dsamp_shape(shape, width=384):
    """img_dsamp_res (compute_branches.py:218-222)"""
    return tuple(int(v) for v in np.multiply(shape[-2:], width / shape[-1]).round().astype(int))


def _as_u16_stack(stack: np.ndarray) -> np.ndarray:
    stack = np.asarray(stack)
    if stack.ndim != 3 or stack.dtype not in (np.uint8, np.uint16):
        raise ValueError("Z stack: expected (Z, H, W) uint8 or uint16")
    return np.ascontiguousarray(stack, np.uint16)


def stack_prepare(handle: Handle, stack: np.ndarray, out_hw) -> np.ndarray:
    """compute_branches.py:247-256 -> (Z, out_h, out_w) float32 in 0..1"""
    a = _as_u16_stack(stack)
    install_gaussian_tables(handle, a.shape[1:], out_hw, sigmas=())
    vol = np.empty((a.shape[0],) + tuple(out_hw), np.float32)
    check(lib().tmat_stack_prepare(handle.raw, ptr(a), a.shape[0], a.shape[1], a.shape[2], out_hw[0], out_hw[1], ptr(vol)), "tmat_stack_prepare")
    return vol


def vessel_field(handle: Handle, vol: np.ndarray, hessian="gaussian_derivatives", return_stages=False):
    """compute_branches.py:258-302: prepared stack (Z, h, w) float32 -> vesselness image (h, w) float32"""
    v = np.ascontiguousarray(vol, np.float32)
    Z, h, w = v.shape
    install_gaussian_tables(handle)
    field = np.empty((h, w), np.float32)
    st = None
    arrays = {}
    if return_stages:
        st = VesselStages()
        for k in ("vess", "sharp"):
            arrays[k] = np.empty((Z - 1, h, w), np.float32)
        arrays["vessels"] = np.empty((h, w), np.float32)
        for k in ("edges", "skel", "mask_sel", "grown", "closed", "filt"):
            arrays[k] = np.empty((h, w), np.uint8)
        for k, a in arrays.items():
            setattr(st, k, a.ctypes.data)
    check(lib().tmat_vessel_field(handle.raw, ptr(v), Z, h, w, HESSIAN_FORMS[hessian], ptr(field), C.byref(st) if st is not None else None),
          "tmat_vessel_field")
    if return_stages:
        return field, {k: (a.astype(bool) if a.dtype == np.uint8 else a) for k, a in arrays.items()}
    return field


def analyze_stack(handle: Handle, stack: np.ndarray, graph_thresh_1, graph_thresh_2, smoothing_window_px, min_branch_length_px,
                  max_branch_length_px=None, remove_isolated=False, ds_width=384, hessian="gaussian_derivatives", index=0, return_field=False):
    """tmat_analyze_stack: one Z stack -> (count, total_px, avg_px[, field])"""
    a = _as_u16_stack(stack)
    out_hw = dsamp_shape(a.shape, ds_width)
    install_gaussian_tables(handle, a.shape[1:], out_hw)
    row = Row()
    field = np.empty(out_hw, np.float32) if return_field else None
    check(lib().tmat_analyze_stack(handle.raw, ptr(a), a.shape[0], a.shape[1], a.shape[2], int(ds_width), HESSIAN_FORMS[hessian], float(graph_thresh_1),
                                   float(graph_thresh_2), int(smoothing_window_px), int(min_branch_length_px), int(max_branch_length_px or 0),
                                   int(bool(remove_isolated)), int(index), C.byref(row), ptr(field) if return_field else None), "tmat_analyze_stack")
    res = (row.count, row.total_px, row.avg_px)
    return res + (field,) if return_field else res


def field_stats(handle: Handle, field: np.ndarray, graph_thresh_1, graph_thresh_2, smoothing_window_px, min_branch_length_px,
                max_branch_length_px=None, remove_isolated=False, index=0, pruning_mask=None):
    """tmat_field_stats(_pruned): vesselness image -> (count, total_px, avg_px); pruning_mask (bool, the field's shape) is MorseGraph's
    `pruning_mask` (compute_branches.py:243, 412-420: the inverse of the shrunken well mask under --detect-well)"""
    f = np.ascontiguousarray(field, np.float32)
    row = Row()
    pm = None
    if pruning_mask is not None:
        pm = np.ascontiguousarray(np.asarray(pruning_mask) > 0, np.uint8)
        if pm.shape != f.shape:
            raise ValueError("field_stats: pruning_mask must have the field's shape")
    check(lib().tmat_field_stats_pruned(handle.raw, ptr(f), f.shape[0], f.shape[1], float(graph_thresh_1), float(graph_thresh_2), int(smoothing_window_px),
                                        int(min_branch_length_px), int(max_branch_length_px or 0), int(bool(remove_isolated)),
                                        ptr(pm) if pm is not None else None, int(index), C.byref(row)), "tmat_field_stats_pruned")
    return row.count, row.total_px, row.avg_px


def resize_aa(handle: Handle, imgs: np.ndarray, out_hw) -> np.ndarray:
    """skimage.transform.resize(x, out_hw, order=1, preserve_range=True, anti_aliasing=True) of one (H, W) or n (n, H, W) uint8 / uint16
    images -> float64 (compute_branches.py:232-238: the max projection of a Z stack before make_well_mask)"""
    a = np.asarray(imgs)
    single = a.ndim == 2
    a = _as_u16_stack(a[None] if single else a)
    install_gaussian_tables(handle, a.shape[1:], out_hw, sigmas=())
    out = np.empty((a.shape[0],) + tuple(int(v) for v in out_hw), np.float64)
    check(lib().tmat_resize_aa_u16(handle.raw, ptr(a), a.shape[0], a.shape[1], a.shape[2], int(out_hw[0]), int(out_hw[1]), ptr(out)), "tmat_resize_aa_u16")
    return out[0] if single else out


def stack_well_masks(handle: Handle, stack: np.ndarray, out_hw, seed=0):
    """--detect-well for a Z stack (compute_branches.py:227-243): well mask of the anti-alias-resized max projection
    -> (well_mask, pruning_mask) bool arrays of shape out_hw (pruning = not shrunken well)"""
    from . import well_mask_generation as wmg
    proj = np.asarray(stack).max(0)
    well, shrunken = wmg.make_well_mask(resize_aa(handle, proj, out_hw), handle=handle, seed=seed)
    return well, np.logical_not(shrunken)


def stack_field(handle: Handle, stack: np.ndarray, ds_width=384, hessian="gaussian_derivatives") -> np.ndarray:
    """compute_branches.py:247-302 for one stack: the (round(H ds_width / W), ds_width) vesselness image"""
    a = _as_u16_stack(stack)
    out_hw = dsamp_shape(a.shape, ds_width)
    return vessel_field(handle, stack_prepare(handle, a, out_hw), hessian)
