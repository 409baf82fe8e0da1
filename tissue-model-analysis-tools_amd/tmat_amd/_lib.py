"""ctypes binding of libtmat_hip.so (include/tmat.h).

The HIP library is the product path.  There is no CPU fallback: if the shared object is missing
or no MI355X is visible, loading / `create()` raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("TMAT_HIP_LIB", _HERE / "libtmat_hip.so"))

_lib = None


class TmatError(RuntimeError):
    pass


class Row(C.Structure):
    _fields_ = [("index", C.c_int64), ("count", C.c_int64), ("total_px", C.c_double), ("avg_px", C.c_double)]


def lib():
    """Load libtmat_hip.so once; raise loudly when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise TmatError(
            f"{LIB_PATH} not found: build it with `python tools/build.py` "
            "(hipcc --offload-arch=gfx950). tmat_amd has no CPU fallback.")
    L = C.CDLL(str(LIB_PATH))
    vp, i, f, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    L.tmat_last_error.restype = C.c_char_p
    L.tmat_version.restype = i
    L.tmat_create.argtypes = [i, vp, sz, i, C.POINTER(vp)]
    L.tmat_create_plain.argtypes = [i, C.POINTER(vp)]
    L.tmat_destroy.argtypes = [vp]
    L.tmat_destroy.restype = None
    L.tmat_sync.argtypes = [vp]
    L.tmat_set_input_depth.argtypes = [vp, i]
    L.tmat_unet_predict.argtypes = [vp, vp, i, vp]
    L.tmat_predict_smooth.argtypes = [vp, vp, i, i, i, vp]
    L.tmat_segment_batch.argtypes = [vp, vp, i, i, i, C.c_double, vp]
    L.tmat_postprocess_batch.argtypes = [vp, vp, i, i, i, i, i, vp]
    L.tmat_filter_edt_batch.argtypes = [vp, vp, i, i, i, vp, vp]
    L.tmat_medial_axis_batch.argtypes = [vp, vp, i, i, i, vp, vp]
    L.tmat_finish_batch.argtypes = [vp, vp, vp, vp, i, i, i, i, i, vp, vp]
    L.tmat_filter_mask_batch.argtypes = [vp, vp, i, i, i, i, i, vp]
    L.tmat_gather_rows.argtypes = [vp, vp, i, vp, vp]
    L.tmat_zproj_batch.argtypes = [vp, vp, i, i, i, i, i, vp]
    L.tmat_zproj_dev.argtypes = [vp, vp, i, i, i, i, i, vp]
    L.tmat_dmt_graph.argtypes = [vp, vp, i, i, f, f, vp, i, vp, i, C.POINTER(i), C.POINTER(i)]
    L.tmat_dmt_graph_batch.argtypes = [vp, vp, i, i, i, f, f, vp, i, vp, i, vp, vp]
    L.tmat_morse_stats.argtypes = [vp, i, vp, i, i, i, i, i, i, i, vp, C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                   C.POINTER(C.c_double), vp, i]
    L.tmat_analyze_batch_dev.argtypes = [vp, vp, i, i, i, C.c_double, i, f, f, i, i, i, i, C.c_int64, vp]
    L.tmat_analyze_batch.argtypes = [vp, vp, i, i, i, C.c_double, i, f, f, i, i, i, i, C.c_int64, vp]
    L.tmat_dev_alloc.argtypes = [vp, sz, C.POINTER(vp)]
    L.tmat_dev_free.argtypes = [vp, vp]
    L.tmat_dev_upload.argtypes = [vp, vp, vp, sz]
    d = C.c_double
    L.tmat_set_gaussian_table.argtypes = [vp, d, i, i, vp]
    L.tmat_host_gaussian_kernel1d.argtypes = [d, i, i, vp]
    L.tmat_gaussian_f32.argtypes = [vp, vp, i, i, i, d, i, vp]
    L.tmat_sato_batch.argtypes = [vp, vp, i, i, i, vp, i, i, vp]
    L.tmat_stack_prepare.argtypes = [vp, vp, i, i, i, i, i, vp]
    L.tmat_vessel_field.argtypes = [vp, vp, i, i, i, i, vp, vp]
    L.tmat_analyze_stack.argtypes = [vp, vp, i, i, i, i, i, f, f, i, i, i, i, C.c_int64, vp, vp]
    L.tmat_field_stats.argtypes = [vp, vp, i, i, f, f, i, i, i, i, C.c_int64, vp]
    L.tmat_field_stats_pruned.argtypes = [vp, vp, i, i, f, f, i, i, i, i, vp, C.c_int64, vp]
    L.tmat_resize_aa_u16.argtypes = [vp, vp, i, i, i, i, i, vp]
    L.tmat_cell_area_batch.argtypes = [vp, vp, i, i, i, i, i, d, vp, vp, vp]
    L.tmat_cell_area_masked.argtypes = [vp, vp, vp, i, i, i, d, vp, vp, vp]
    L.tmat_resize_linear_u16.argtypes = [vp, vp, i, i, i, i, i, vp]
    L.tmat_resnet_load.argtypes = [vp, vp, sz, C.POINTER(i)]
    L.tmat_resnet_predict.argtypes = [vp, i, vp, i, i, vp]
    L.tmat_inv_depth_predict.argtypes = [vp, vp, i, vp, i, i, i, i, vp, vp]
    L.tmat_inv_depth_predict_multi.argtypes = [vp, vp, i, vp, vp, i, i, i, i, vp]
    L.tmat_prof_enable.argtypes = [vp, i]
    L.tmat_debug_poison.argtypes = [vp, i]
    L.tmat_set_precision.argtypes = [vp, i]
    L.tmat_set_input_norm.argtypes = [vp, i, C.c_double, C.c_double]
    L.tmat_preprocess_batch.argtypes = [vp, vp, i, i, i, C.c_double, vp]
    L.tmat_well_threshold.argtypes = [vp, vp, i, i, vp]
    L.tmat_well_threshold_f64.argtypes = [vp, vp, i, i, vp]
    L.tmat_canny_mask.argtypes = [vp, vp, i, i, C.c_double, vp]
    L.tmat_prof_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double), i]
    for name in EXPORTS:
        fn = getattr(L, name)
        if name not in ("tmat_last_error", "tmat_destroy"):
            fn.restype = i
    _lib = L
    return L


EXPORTS = [
    "tmat_last_error", "tmat_version", "tmat_create", "tmat_create_plain", "tmat_destroy", "tmat_sync", "tmat_set_input_depth", "tmat_unet_predict",
    "tmat_predict_smooth", "tmat_segment_batch", "tmat_postprocess_batch", "tmat_filter_edt_batch", "tmat_medial_axis_batch", "tmat_finish_batch", "tmat_filter_mask_batch", "tmat_zproj_batch", "tmat_zproj_dev", "tmat_gather_rows",
    "tmat_dmt_graph", "tmat_dmt_graph_batch", "tmat_morse_stats",
    "tmat_analyze_batch_dev", "tmat_analyze_batch", "tmat_dev_alloc", "tmat_dev_free", "tmat_dev_upload",
    "tmat_prof_enable", "tmat_prof_read", "tmat_debug_poison", "tmat_set_precision", "tmat_set_input_norm", "tmat_preprocess_batch", "tmat_well_threshold", "tmat_well_threshold_f64", "tmat_canny_mask", "tmat_host_lanczos4_u16", "tmat_host_rescale01_u16",
    "tmat_host_rescale255_f32", "tmat_host_filter_mask", "tmat_host_skeletonize", "tmat_host_medial_axis",
    "tmat_host_permutation", "tmat_host_postprocess",
    "tmat_set_gaussian_table", "tmat_host_gaussian_kernel1d", "tmat_gaussian_f32", "tmat_sato_batch", "tmat_stack_prepare", "tmat_vessel_field",
    "tmat_analyze_stack", "tmat_field_stats", "tmat_field_stats_pruned", "tmat_resize_aa_u16", "tmat_cell_area_batch", "tmat_cell_area_masked", "tmat_resize_linear_u16",
    "tmat_resnet_load", "tmat_resnet_predict", "tmat_inv_depth_predict", "tmat_inv_depth_predict_multi",
]


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().tmat_last_error()
        raise TmatError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class VesselStages(C.Structure):
    """tmat_vessel_stages (include/tmat.h)"""
    _fields_ = [(k, C.c_void_p) for k in ("vess", "sharp", "vessels", "edges", "skel", "mask_sel", "grown", "closed", "filt")]


class Handle:
    """Owns one tmat_handle (one HIP device + stream + resident weights)."""

    def __init__(self, weights_blob: "bytes | None", device_id: int = 0, max_patches: int = 0):
        """weights_blob None: a handle without a model (tmat_create_plain), enough for zproj / filter_edt / finish"""
        L = lib()
        self._h = C.c_void_p()
        self._blob = weights_blob
        if weights_blob is None:
            check(L.tmat_create_plain(device_id, C.byref(self._h)), "tmat_create_plain")
            return
        buf = (C.c_char * len(weights_blob)).from_buffer_copy(weights_blob)
        check(L.tmat_create(device_id, C.cast(buf, C.c_void_p), len(weights_blob), max_patches, C.byref(self._h)),
              "tmat_create")

    @property
    def raw(self):
        return self._h

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().tmat_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- stage entry points --------------------------------------------------------------
    def unet_predict(self, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float32)
        y = np.empty_like(x)
        check(lib().tmat_unet_predict(self._h, ptr(x), x.shape[0], ptr(y)), "tmat_unet_predict")
        return y

    def predict_smooth(self, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float32)
        single = x.ndim == 2
        xb = x[None] if single else x
        out = np.empty(xb.shape, np.float64)
        check(lib().tmat_predict_smooth(self._h, ptr(xb), xb.shape[0], xb.shape[1], xb.shape[2], ptr(out)),
              "tmat_predict_smooth")
        return out[0] if single else out

    def medial_axis(self, mask: np.ndarray):
        """tmat_medial_axis_batch: mask (n, h, w) bool/u8 -> (skel (n, h, w) bool, dist (n, h, w) f64), on the device"""
        m = np.ascontiguousarray(mask, np.uint8)
        skel = np.empty(m.shape, np.uint8)
        dist = np.empty(m.shape, np.float64)
        check(lib().tmat_medial_axis_batch(self._h, ptr(m), m.shape[0], m.shape[1], m.shape[2], ptr(skel), ptr(dist)), "tmat_medial_axis_batch")
        return skel.astype(bool), dist

    def filter_edt(self, pred: np.ndarray):
        """GPU binary morphology: pred (n, h, w) f64 -> (filtered mask bool, EDT f64)"""
        pred = np.ascontiguousarray(pred, np.float64)
        filt = np.empty(pred.shape, np.uint8)
        dist = np.empty(pred.shape, np.float64)
        check(lib().tmat_filter_edt_batch(self._h, ptr(pred), pred.shape[0], pred.shape[1], pred.shape[2], ptr(filt), ptr(dist)),
              "tmat_filter_edt_batch")
        return filt.astype(bool), dist

    def finish(self, pred, dist, skel, out_shape):
        """GPU: EDT(~skel), centre-line weighting, anti-aliased resize, rescale -> (field f32, field255 f32)"""
        pred = np.ascontiguousarray(pred, np.float64)
        dist = np.ascontiguousarray(dist, np.float64)
        skel = np.ascontiguousarray(np.asarray(skel) != 0, np.uint8)
        n = pred.shape[0]
        f = np.empty((n,) + tuple(out_shape), np.float32)
        f255 = np.empty_like(f)
        check(lib().tmat_finish_batch(self._h, ptr(pred), ptr(dist), ptr(skel), n, pred.shape[1], pred.shape[2], out_shape[0],
                                      out_shape[1], ptr(f), ptr(f255)), "tmat_finish_batch")
        return f, f255

    def filter_mask(self, masks, use_median=True, remove_isolated=True):
        """GPU filter_branch_seg_mask on (n, h, w) masks -> bool (n, h, w)"""
        m = np.ascontiguousarray(np.asarray(masks) != 0, np.uint8)
        out = np.empty_like(m)
        check(lib().tmat_filter_mask_batch(self._h, ptr(m), m.shape[0], m.shape[1], m.shape[2], int(bool(use_median)),
                                           int(bool(remove_isolated)), ptr(out)), "tmat_filter_mask_batch")
        return out.astype(bool)

    ZPROJ_METHODS = {"fs": 0, "min": 1, "max": 2, "avg": 3, "med": 4}

    def zproj(self, stacks, method="fs"):
        """GPU Z projection of (n, Z, H, W) uint8/uint16 stacks -> (n, H, W); dtype as zstacks.py returns it
        (input dtype for fs / min / max, float64 for avg / med)"""
        stacks = np.asarray(stacks)
        if stacks.ndim != 4 or stacks.dtype not in (np.uint8, np.uint16):
            raise ValueError("zproj: expected (n, Z, H, W) uint8 or uint16 stacks")
        m = self.ZPROJ_METHODS[method]
        a = np.ascontiguousarray(stacks, np.uint16)
        n, Z, H, W = a.shape
        out = np.empty((n, H, W), np.float64 if m >= 3 else np.uint16)
        check(lib().tmat_zproj_batch(self._h, ptr(a), n, Z, H, W, m, ptr(out)), "tmat_zproj_batch")
        return out.astype(stacks.dtype) if m < 3 else out

    def set_input_norm(self, norm_mean=None, norm_std=None):
        """models.py:636-637 on the device: x = (x - norm_mean) / norm_std in front of the smooth prediction; None turns it off"""
        on = norm_mean is not None and norm_std is not None
        check(lib().tmat_set_input_norm(self._h, int(on), float(norm_mean or 0.0), float(norm_std if on else 1.0)), "tmat_set_input_norm")

    def set_precision(self, mode="f32"):
        """arithmetic of the UNet's dense convolutions: "f32" (bit-exact contract, default) "bf16x3" or "bf16x6" (opt-in split
        precision on the bf16 matrix cores, include/tmat.h:tmat_set_precision)"""
        modes = {"f32": 0, "bf16x3": 1, "bf16x6": 2}
        if mode not in modes:
            raise ValueError(f"precision must be one of {sorted(modes)}")
        check(lib().tmat_set_precision(self._h, modes[mode]), "tmat_set_precision")

    def debug_poison(self, byte_pattern=0xFF):
        """test-only: fill every scratch workspace of the handle with a byte pattern (include/tmat.h:tmat_debug_poison)"""
        check(lib().tmat_debug_poison(self._h, int(byte_pattern)), "tmat_debug_poison")

    def prof_enable(self, on=True):
        check(lib().tmat_prof_enable(self._h, int(on)), "tmat_prof_enable")

    def prof_read(self, reset=True):
        ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
        check(lib().tmat_prof_read(self._h, C.byref(ms), C.byref(n), C.byref(fl), int(reset)), "tmat_prof_read")
        return ms.value, n.value, fl.value


# -- host-only entry points (no handle / no GPU needed) ------------------------------------------
def dmt_graph(img: np.ndarray, delta1: float, delta2: float = 0.0, handle: "Handle | None" = None):
    """tmat_dmt_graph: (vertices (n,2) int32 [row, col], edges (m,2) int32)."""
    img = np.ascontiguousarray(img, np.float32)
    R, Cc = img.shape
    cap_v, cap_e = R * Cc + 4, 3 * R * Cc + 4
    V = np.empty((cap_v, 2), np.int32)
    E = np.empty((cap_e, 2), np.int32)
    nv, ne = C.c_int(), C.c_int()
    check(lib().tmat_dmt_graph(handle.raw if handle else None, ptr(img), R, Cc, float(delta1), float(delta2), ptr(V), cap_v,
                               ptr(E), cap_e, C.byref(nv), C.byref(ne)), "tmat_dmt_graph")
    return V[: nv.value].copy(), E[: ne.value].copy()


def dmt_graph_batch(imgs: np.ndarray, delta1: float, delta2: float = 0.0, handle: "Handle | None" = None):
    """tmat_dmt_graph_batch: n fields of one shape in one call -> list of (vertices, edges) as dmt_graph returns them."""
    imgs = np.ascontiguousarray(imgs, np.float32)
    n, R, Cc = imgs.shape
    cap_v, cap_e = R * Cc + 4, 3 * R * Cc + 4
    V = np.empty((n, cap_v, 2), np.int32)
    E = np.empty((n, cap_e, 2), np.int32)
    nv, ne = np.zeros(n, np.int32), np.zeros(n, np.int32)
    check(lib().tmat_dmt_graph_batch(handle.raw if handle else None, ptr(imgs), n, R, Cc, float(delta1), float(delta2), ptr(V), cap_v,
                                     ptr(E), cap_e, ptr(nv), ptr(ne)), "tmat_dmt_graph_batch")
    return [(V[k, : nv[k]].copy(), E[k, : ne[k]].copy()) for k in range(n)]


def morse_stats(V, E, shape, smoothing_window, min_branch_length, max_branch_length=None,
                remove_isolated_branches=False, pruning_mask=None):
    """tmat_morse_stats: (bars (k,2) f64, count, total_px, avg_px)."""
    V = np.ascontiguousarray(V, np.int32).reshape(-1, 2)
    E = np.ascontiguousarray(E, np.int32).reshape(-1, 2)
    pm = None
    if pruning_mask is not None:
        pm = np.ascontiguousarray(np.asarray(pruning_mask) > 0, np.uint8)
    cap = max(len(V), 1)
    bars = np.empty((cap, 2), np.float64)
    cnt, tot, avg = C.c_int64(), C.c_double(), C.c_double()
    check(lib().tmat_morse_stats(ptr(V), len(V), ptr(E), len(E), int(shape[0]), int(shape[1]), int(smoothing_window),
                                 int(min_branch_length), int(max_branch_length or 0), int(bool(remove_isolated_branches)),
                                 ptr(pm) if pm is not None else None, C.byref(cnt), C.byref(tot), C.byref(avg), ptr(bars), cap),
          "tmat_morse_stats")
    return bars[: cnt.value].copy(), cnt.value, tot.value, avg.value


# -- host pixel stages (csrc/postproc.cpp); exposed for stage-wise parity tests --------------------
def host_lanczos4_u16(img, out_hw):
    img = np.ascontiguousarray(img, np.uint16)
    out = np.empty(out_hw, np.uint16)
    L = lib()
    L.tmat_host_lanczos4_u16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    check(L.tmat_host_lanczos4_u16(ptr(img), img.shape[0], img.shape[1], out_hw[0], out_hw[1], ptr(out)), "lanczos4")
    return out


def host_rescale01_u16(img):
    img = np.ascontiguousarray(img, np.uint16)
    out = np.empty(img.shape, np.float32)
    L = lib()
    L.tmat_host_rescale01_u16.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    check(L.tmat_host_rescale01_u16(ptr(img), img.size, ptr(out)), "rescale01")
    return out


def host_rescale255_f32(img):
    img = np.ascontiguousarray(img, np.float32)
    out = np.empty(img.shape, np.float32)
    L = lib()
    L.tmat_host_rescale255_f32.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    check(L.tmat_host_rescale255_f32(ptr(img), img.size, ptr(out)), "rescale255")
    return out


def host_filter_mask(mask, use_median=True, remove_isolated=True):
    m = np.ascontiguousarray(np.asarray(mask) != 0, np.uint8)
    out = np.empty(m.shape, np.uint8)
    L = lib()
    L.tmat_host_filter_mask.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    check(L.tmat_host_filter_mask(ptr(m), m.shape[0], m.shape[1], int(use_median), int(remove_isolated), ptr(out)), "filter")
    return out.astype(bool)


def host_skeletonize(mask):
    m = np.ascontiguousarray(np.asarray(mask) != 0, np.uint8)
    out = np.empty(m.shape, np.uint8)
    L = lib()
    L.tmat_host_skeletonize.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    check(L.tmat_host_skeletonize(ptr(m), m.shape[0], m.shape[1], ptr(out)), "skeletonize")
    return out.astype(bool)


def host_medial_axis(mask):
    m = np.ascontiguousarray(np.asarray(mask) != 0, np.uint8)
    sk = np.empty(m.shape, np.uint8)
    dist = np.empty(m.shape, np.float64)
    L = lib()
    L.tmat_host_medial_axis.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    check(L.tmat_host_medial_axis(ptr(m), m.shape[0], m.shape[1], ptr(sk), ptr(dist)), "medial_axis")
    return sk.astype(bool), dist


def host_permutation(seed, n):
    out = np.empty(n, np.uint32)
    L = lib()
    L.tmat_host_permutation.argtypes = [C.c_uint32, C.c_int, C.c_void_p]
    check(L.tmat_host_permutation(seed, n, ptr(out)), "permutation")
    return out


def host_postprocess(pred, out_shape):
    pred = np.ascontiguousarray(pred, np.float64)
    out = np.empty(out_shape, np.float32)
    L = lib()
    L.tmat_host_postprocess.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    check(L.tmat_host_postprocess(ptr(pred), pred.shape[0], pred.shape[1], out_shape[0], out_shape[1], ptr(out)), "postprocess")
    return out
