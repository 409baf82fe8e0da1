"""Well detection (`--detect-well`): the reference's fl_tissue_model_tools/well_mask_generation.py (same function names and
arguments) over the C-ABI.

The pixel stages run on the GPU -- `auto_threshold_well` (gaussian, uint8 rescale, corner medians, Otsu threshold, disk(5)
erosion) is one call, tmat_well_threshold, and the two `skimage.feature.canny` calls are tmat_canny_mask.  Host code keeps
what the reference does on a <= 200-pixel image with library calls: the nearest-neighbour rescale (index arithmetic),
scipy.spatial.ConvexHull (the reference's own dependency), the hull mask (exact integer point-in-polygon test in place of
Delaunay.find_simplex), and the random superellipse search (:16-91), restated line by line with numpy.

The reference draws its 25 000 candidates from the GLOBAL numpy generator and never seeds it (:35), so two runs of the
reference give different masks.  Here the draw is `numpy.random.RandomState(seed).rand(num_iters, 6)` with an explicit `seed`
(scripts: --well-seed, default 0): the same stream the reference consumes after `numpy.random.seed(seed)`.

scikit-image version note (reference pins 0.22.0): `rescale` / `resize` with order 0 follow scipy.ndimage.zoom(order=0,
grid_mode=True), i.e. source index floor((i + 0.5) * n_in / n_out); canny follows the 0.18.3 source (magnitude by hypot).
"""
from __future__ import annotations

from numbers import Integral

import numpy as np

from . import _lib
from .sato import ensure_gaussian_table

SUPERELLIPSE_BOUNDS = np.array([
    (-np.pi / 20, np.pi / 20),   # theta
    (0.67, 1.33),                # d
    (0.9, 1.1),                  # s_a
    (0.9, 1.1),                  # s_b
    (-0.3, 0.3),                 # c_x
    (-0.3, 0.3),                 # c_y
])


def _gamma(x: float) -> float:
    from scipy.special import gamma          # the reference's own call (:10, :79-82)
    return gamma(x)


def get_superellipse_hull(x, y, n, num_iters=25000, seed=0):
    """Find a superellipse that encloses the given points (reference :16-91).  Returns (t, d, s_a, s_b, c_x, c_y); raises
    ValueError when no candidate encloses them (the reference's np.argmin of an empty sequence)."""
    linear_weights = np.random.RandomState(seed).rand(num_iters, 6)
    param_values = (SUPERELLIPSE_BOUNDS[:, 1] - SUPERELLIPSE_BOUNDS[:, 0]) * linear_weights + SUPERELLIPSE_BOUNDS[:, 0]
    t, d, s_a, s_b, c_x, c_y = param_values.T[..., np.newaxis]
    if n == 2:
        val = ((x - c_x) / (d * s_a)) ** 2 + ((y - c_y) / (d * s_b)) ** 2
    elif n % 2 == 0:
        val = ((((x - c_x) * np.cos(t) - ((y - c_y) * np.sin(t))) / (d * s_a)) ** n
               + (((x - c_x) * np.sin(t) + (y - c_y) * np.cos(t)) / (d * s_b)) ** n)
    else:
        val = (np.abs(((x - c_x) * np.cos(t) - ((y - c_y) * np.sin(t))) / (d * s_a)) ** n
               + np.abs(((x - c_x) * np.sin(t) + (y - c_y) * np.cos(t)) / (d * s_b)) ** n)
    candidate_indices = np.where(np.max(val, axis=1) < 1)[0]
    t, d, s_a, s_b, c_x, c_y = (q[candidate_indices] for q in (t, d, s_a, s_b, c_x, c_y))
    smallest_area_idx = np.argmin(4 * d ** 2 * s_a * s_b * _gamma(1 + 1 / n) ** 2 / _gamma(1 + 2 / n))
    return tuple(q[smallest_area_idx][0] for q in (t, d, s_a, s_b, c_x, c_y))


def gen_superellipse_mask(t, d, s_a, s_b, c_x, c_y, n, shape) -> np.ndarray:
    """reference :94-118"""
    x = np.linspace(-1, 1, shape[0])
    y = np.linspace(-1, 1, shape[1])
    X, Y = np.meshgrid(x, y)
    mask = ((np.abs(((X - c_x) * np.cos(t) - (Y - c_y) * np.sin(t)) / (d * s_a))) ** n
            + (np.abs(((X - c_x) * np.sin(t) + (Y - c_y) * np.cos(t)) / (d * s_b))) ** n
            < 1)
    return np.swapaxes(mask, 0, 1)


def create_convex_hull_mask(array_shape, hull_vertices) -> np.ndarray:
    """reference :121-139 (Delaunay(hull_vertices).find_simplex(all pixels) >= 0): the pixels inside or on the convex
    polygon, by exact integer cross products against every edge"""
    v = np.asarray(hull_vertices, np.int64)
    rr, cc = np.indices(array_shape)
    k = len(v)
    area2 = sum(int(v[i][0]) * int(v[(i + 1) % k][1]) - int(v[(i + 1) % k][0]) * int(v[i][1]) for i in range(k))
    sign = 1 if area2 > 0 else -1
    mask = np.ones(array_shape, bool)
    for i in range(k):
        a, b = v[i], v[(i + 1) % k]
        mask &= ((b[0] - a[0]) * (cc - a[1]) - (b[1] - a[1]) * (rr - a[0])) * sign >= 0
    return mask


def _resize_nearest(a: np.ndarray, shape) -> np.ndarray:
    """skimage.transform.resize(a, shape, order=0, preserve_range=True): source index floor((i + 0.5) * n_in / n_out)"""
    a = np.asarray(a)
    idx = [np.minimum(np.floor((np.arange(o) + 0.5) * (i / o)).astype(np.int64), i - 1) for o, i in zip(shape, a.shape)]
    return a[np.ix_(idx[0], idx[1])]


def _border(handle, mask: np.ndarray) -> np.ndarray:
    """canny(mask) plus the mask's pixels on the image border (reference :165-170, :201-205)"""
    m = np.ascontiguousarray(np.asarray(mask) != 0, np.uint8)
    ensure_gaussian_table(handle, 1.0, 0, 4.0)
    edges = np.empty_like(m)
    _lib.check(_lib.lib().tmat_canny_mask(handle.raw, _lib.ptr(m), m.shape[0], m.shape[1], 1.0, _lib.ptr(edges)), "tmat_canny_mask")
    b = edges.astype(bool)
    mb = m.astype(bool)
    b[0, :] |= mb[0, :]; b[-1, :] |= mb[-1, :]; b[:, 0] |= mb[:, 0]; b[:, -1] |= mb[:, -1]
    return b


def auto_threshold_well(image: np.ndarray, handle: _lib.Handle) -> np.ndarray:
    """Threshold an image to get a rough mask of the well (reference :236-277), on the GPU"""
    image = np.asarray(image)
    if image.ndim != 2:
        raise ValueError("auto_threshold_well: 2-D image expected")
    ensure_gaussian_table(handle, 1.0, 0, 4.0)
    out = np.empty(image.shape, np.uint8)
    if image.dtype == np.float32:
        img = np.ascontiguousarray(image)
        _lib.check(_lib.lib().tmat_well_threshold(handle.raw, _lib.ptr(img), img.shape[0], img.shape[1], _lib.ptr(out)), "tmat_well_threshold")
    else:
        # skimage's gaussian() converts with img_as_float: unsigned integers are multiplied by 1 / max in float64
        if image.dtype.kind == "u":
            img = np.multiply(image, 1.0 / np.iinfo(image.dtype).max, dtype=np.float64)
        elif image.dtype == np.float64:
            img = np.ascontiguousarray(image)
        else:
            raise ValueError("auto_threshold_well: float32 / float64 / unsigned integer images")
        _lib.check(_lib.lib().tmat_well_threshold_f64(handle.raw, _lib.ptr(img), img.shape[0], img.shape[1], _lib.ptr(out)), "tmat_well_threshold_f64")
    return out.astype(bool)


def generate_well_mask(image: np.ndarray, mask_val: Integral = 1, return_superellipse_params: bool = False, *,
                       handle: _lib.Handle, seed: int = 0):
    """Generate a binary mask over the well in an image (reference :142-233)."""
    from scipy.spatial import ConvexHull
    try:
        from scipy.spatial import QhullError
    except ImportError:                                     # older scipy
        from scipy.spatial.qhull import QhullError
    image = np.asarray(image)
    im_thresh = auto_threshold_well(image, handle)
    downsamp_ratio = min(1, 200 / np.max(im_thresh.shape))
    small_shape = tuple(int(v) for v in np.round(np.asarray(im_thresh.shape) * downsamp_ratio))
    im_thresh = _resize_nearest(im_thresh, small_shape)
    border_points = np.argwhere(_border(handle, im_thresh))

    def get_circ_mask():
        circ_mask = np.zeros(image.shape, dtype=np.uint8)
        center = image.shape[0] // 2, image.shape[1] // 2
        radius = int(image.shape[0] * 0.5 * (1 - 0.95))
        rr, cc = np.indices(image.shape)
        circ_mask[(rr - center[0]) ** 2 + (cc - center[1]) ** 2 < radius ** 2] = mask_val      # skimage.draw.disk
        return circ_mask

    try:
        hull = ConvexHull(border_points)
    except (ValueError, QhullError):                        # QhullError derives from ValueError in the pinned scipy
        return get_circ_mask()
    hull_vertices = border_points[hull.vertices]
    well_mask = create_convex_hull_mask(im_thresh.shape, hull_vertices)
    well_mask_border = _border(handle, well_mask)
    area = np.sum(well_mask)
    perimeter = np.sum(well_mask_border)
    n = 8 if perimeter / area > .027 else 2
    x = hull_vertices[:, 0] / im_thresh.shape[0] * 2 - 1
    y = hull_vertices[:, 1] / im_thresh.shape[1] * 2 - 1
    found_superellipse = False
    try:
        t, d, s_a, s_b, c_x, c_y = get_superellipse_hull(x, y, n, seed=seed)
        d *= 0.9
        well_mask = gen_superellipse_mask(t, d, s_a, s_b, c_x, c_y, n, im_thresh.shape)
        found_superellipse = True
    except ValueError:
        print("Falling back to convex hull well mask.", flush=True)
    well_mask = well_mask.astype(np.uint8) * mask_val
    well_mask = _resize_nearest(well_mask, image.shape).astype(np.float64)          # skimage's resize returns floats
    if found_superellipse and return_superellipse_params:
        return well_mask, t, d, s_a, s_b, c_x, c_y, n
    return well_mask


def make_well_mask(img: np.ndarray, *, handle: _lib.Handle, seed: int = 0, warn=print):
    """scripts/compute_branches.py:109-141: (well_mask, shrunken_well_mask) as boolean arrays; the second one (a 10 % smaller
    superellipse, or the mask eroded by disk(5) when the fit failed) becomes the pruning mask"""
    res = generate_well_mask(img, return_superellipse_params=True, handle=handle, seed=seed)
    if isinstance(res, tuple):
        well_mask, t, d, s_a, s_b, c_x, c_y, n = res
        well_mask = well_mask > 0
        d *= 0.9
        shrunken_well_mask = gen_superellipse_mask(t, d, s_a, s_b, c_x, c_y, n, img.shape[:2])
    else:
        well_mask = res > 0
        shrunken_well_mask = _erode_disk5(well_mask)
    coverage = np.sum(well_mask) / well_mask.size
    if coverage < 0.4:
        warn(f"Well mask coverage is too low ({coverage * 100:.2f}%) so it will not be used for analysis.")
        well_mask = np.full(img.shape, True, dtype=bool)
        shrunken_well_mask = np.full(img.shape, True, dtype=bool)
    return well_mask, shrunken_well_mask


def _erode_disk5(mask: np.ndarray) -> np.ndarray:
    """skimage binary_erosion(mask, disk(5)) (border counts as set) of the fallback mask: 81 shifted ANDs"""
    m = np.asarray(mask, bool)
    H, W = m.shape
    pad = np.ones((H + 10, W + 10), bool)
    pad[5:5 + H, 5:5 + W] = m
    out = np.ones((H, W), bool)
    for dy in range(-5, 6):
        for dx in range(-5, 6):
            if dy * dy + dx * dx <= 25:
                out &= pad[5 + dy:5 + dy + H, 5 + dx:5 + dx + W]
    return out
