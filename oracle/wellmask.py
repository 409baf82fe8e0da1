"""oracle/wellmask.py -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU restatement of the reference's well detection, fl_tissue_model_tools/well_mask_generation.py:16-277
(`generate_well_mask`, `auto_threshold_well`, `get_superellipse_hull`, `gen_superellipse_mask`,
`create_convex_hull_mask`) and of scripts/compute_branches.py:109-141 (`make_well_mask`), with numpy / scipy only.

The reference's arithmetic here is scikit-image's (setup.py pins 0.22.0; only 0.18.3 is importable, under /opt/conda):
  gaussian -> scipy.ndimage.gaussian_filter(mode="nearest", truncate 4) on the float32 image
  rescale_intensity(out_range=(0, 255)) -> exposure.py: ((x - min) / (max - min)) * 255 in the image's float32, then uint8
  threshold_otsu -> filters/thresholding.py: bincount histogram between the image's extrema, f64 cumulative sums
  binary_erosion(footprint=disk(5)) -> scipy.ndimage.binary_erosion(border_value=True)
  rescale / resize(order=0) -> scipy.ndimage.zoom(order=0, grid_mode=True) (0.19+ semantics, [recalled]; keeps bool)
  canny(sigma=1) -> feature/_canny.py (0.18.3): gaussian(mode="constant") of image and of the all-ones mask, quotient, then the
                    sobel / non-maximum-suppression / hysteresis steps of oracle/sato.py:canny_from_smoothed
  ConvexHull, Delaunay.find_simplex -> scipy.spatial (qhull); the hull mask is restated as an exact integer
                    point-in-convex-polygon test and checked against Delaunay in tests/test_oracle_wellmask.py
PINNED by tests/golden/wellmask.npz (tools/make_goldens.py wellmask: the reference module imported under scikit-image
0.18.3 with the three bridges described there): thresholded mask, final mask and superellipse parameters, 14 cases (7 inputs x 2 seeds).
The random superellipse search is unseeded in the reference (np.random.rand, :35); here the seed is explicit and the
draw is RandomState(seed).rand(25000, 6) -- the same stream as np.random.seed(seed) followed by np.random.rand.
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage as ndi
from scipy.spatial import ConvexHull
from scipy.special import gamma

from . import sato as osato


def disk(radius: int) -> np.ndarray:
    L = np.arange(-radius, radius + 1)
    X, Y = np.meshgrid(L, L)
    return (X ** 2 + Y ** 2 <= radius ** 2).astype(np.uint8)


def img_as_float(image: np.ndarray) -> np.ndarray:
    """skimage.util.img_as_float as filters.gaussian applies it (util/dtype.py:_convert): floats pass, unsigned integers are
    MULTIPLIED by 1 / max in float64"""
    image = np.asarray(image)
    if image.dtype in (np.float32, np.float64):
        return image
    if image.dtype == bool:
        return image.astype(np.float64)
    if image.dtype.kind == "u":
        return np.multiply(image, 1.0 / np.iinfo(image.dtype).max, dtype=np.float64)
    raise ValueError("img_as_float: unsigned integer or float images only")


def rescale_0_255_u8(im: np.ndarray) -> np.ndarray:
    """rescale_intensity(im, out_range=(0, 255)).astype(uint8) for a float32 image (exposure.py:404-428 of 0.18.3)"""
    imin, imax = float(im.min()), float(im.max())
    im = np.clip(im, imin, imax)
    if imin != imax:
        im = (im - np.float32(imin)) / np.float32(imax - imin) if im.dtype == np.float32 else (im - imin) / (imax - imin)
        return np.asarray(im * 255.0 + 0.0, dtype=np.float64).astype(np.uint8)
    return np.clip(im, 0.0, 255.0).astype(np.float64).astype(np.uint8)


def threshold_otsu_u8(im: np.ndarray, counts_dtype=np.float64):
    """filters/thresholding.py:threshold_otsu on an integer image.  counts_dtype: scikit-image 0.18.3 (the version the goldens were
    generated with, and what the device evaluates) turns the histogram counts into float64; from 0.19 on -- the reference pins 0.22.0 --
    _validate_image_histogram casts them to float32 before the cumulative sums (recalled from the published source: 0.22 is not
    importable here).  Passing np.float32 evaluates that form; tests/test_oracle_wellmask.py asserts both pick the same threshold on
    every fixture case (a near-tie of the between-class variance could separate them: parity against 0.22 is unpinned there)."""
    first = im.ravel()[0]
    if np.all(im == first):
        return first
    lo, hi = int(im.min()), int(im.max())
    counts = np.bincount(im.ravel(), minlength=hi + 1)[lo:hi + 1].astype(counts_dtype)
    centers = np.arange(lo, hi + 1)
    w1 = np.cumsum(counts)
    w2 = np.cumsum(counts[::-1])[::-1]
    m1 = np.cumsum(counts * centers) / w1
    m2 = (np.cumsum((counts * centers)[::-1]) / w2[::-1])[::-1]
    var12 = w1[:-1] * w2[1:] * (m1[:-1] - m2[1:]) ** 2
    return centers[int(np.argmax(var12))]


def auto_threshold_well(image: np.ndarray) -> np.ndarray:
    """well_mask_generation.py:236-277"""
    image = np.asarray(image)
    fimg = img_as_float(image)
    im_blur = ndi.gaussian_filter(fimg, 1, mode="nearest", truncate=4.0)
    im_blur = rescale_0_255_u8(im_blur)
    ext = int(im_blur.min()), int(im_blur.max())
    xl, xr = int(image.shape[0] * 0.05), int(image.shape[0] * 0.95)
    yt, yb = int(image.shape[1] * 0.05), int(image.shape[1] * 0.95)
    meds = [np.median(im_blur[:xl, :yt]), np.median(im_blur[:xl, yb:]), np.median(im_blur[xr:, :yt]), np.median(im_blur[xr:, yb:])]
    if np.abs(ext[0] - min(meds)) > np.abs(ext[1] - max(meds)):
        im_blur = 255 - im_blur
    thresh = threshold_otsu_u8(im_blur)
    return ndi.binary_erosion(im_blur >= thresh, structure=disk(5), border_value=True)


def resize_nearest(a: np.ndarray, shape) -> np.ndarray:
    """skimage.transform.resize(a, shape, order=0, preserve_range=True) from 0.19 on: ndi.zoom on the pixel-centre grid"""
    a = np.asarray(a)
    zoom = [o / i for o, i in zip(shape, a.shape)]
    out = ndi.zoom(a.astype(np.float64), zoom, order=0, mode="nearest", grid_mode=True)
    return out.astype(bool) if a.dtype == bool else out


def canny(image: np.ndarray, sigma: float = 1.0) -> np.ndarray:
    """skimage.feature.canny(image, sigma) of a boolean / float image, default thresholds, no mask (0.18.3 _canny.py)"""
    img = np.asarray(image).astype(np.float64)
    sm = ndi.gaussian_filter(img, sigma, mode="constant", cval=0, truncate=4.0)
    bleed = ndi.gaussian_filter(np.ones(img.shape), sigma, mode="constant", cval=0, truncate=4.0)
    return osato.canny_from_smoothed(sm / (bleed + np.finfo(float).eps))


def border_of(mask: np.ndarray) -> np.ndarray:
    """canny(mask) plus the mask's own pixels on the image border (well_mask_generation.py:165-170, 201-205)"""
    b = canny(mask)
    m = np.asarray(mask).astype(bool)
    b[0, :] |= m[0, :]; b[-1, :] |= m[-1, :]; b[:, 0] |= m[:, 0]; b[:, -1] |= m[:, -1]
    return b


def convex_hull_mask(shape, hull_vertices: np.ndarray) -> np.ndarray:
    """create_convex_hull_mask (:121-139): pixels inside or on the hull.  hull_vertices in counter-clockwise order
    (ConvexHull.vertices of 2-D points); exact integer cross products"""
    v = np.asarray(hull_vertices, np.int64)
    rr, cc = np.indices(shape)
    inside = np.ones(shape, bool)
    sign = None
    for k in range(len(v)):
        a, b = v[k], v[(k + 1) % len(v)]
        cr = (b[0] - a[0]) * (cc - a[1]) - (b[1] - a[1]) * (rr - a[0])
        if sign is None:
            # orientation from the polygon's signed area
            area2 = sum(int(v[i][0]) * int(v[(i + 1) % len(v)][1]) - int(v[(i + 1) % len(v)][0]) * int(v[i][1]) for i in range(len(v)))
            sign = 1 if area2 > 0 else -1
        inside &= (cr * sign >= 0)
    return inside


SUPERELLIPSE_BOUNDS = np.array([(-np.pi / 20, np.pi / 20), (0.67, 1.33), (0.9, 1.1), (0.9, 1.1), (-0.3, 0.3), (-0.3, 0.3)])


def get_superellipse_hull(x, y, n, seed, num_iters=25000):
    """well_mask_generation.py:16-91 with the random draw made explicit"""
    w = np.random.RandomState(seed).rand(num_iters, 6)
    pv = (SUPERELLIPSE_BOUNDS[:, 1] - SUPERELLIPSE_BOUNDS[:, 0]) * w + SUPERELLIPSE_BOUNDS[:, 0]
    t, d, s_a, s_b, c_x, c_y = pv.T[..., np.newaxis]
    if n == 2:
        val = ((x - c_x) / (d * s_a)) ** 2 + ((y - c_y) / (d * s_b)) ** 2
    elif n % 2 == 0:
        val = ((((x - c_x) * np.cos(t) - ((y - c_y) * np.sin(t))) / (d * s_a)) ** n
               + (((x - c_x) * np.sin(t) + (y - c_y) * np.cos(t)) / (d * s_b)) ** n)
    else:
        val = (np.abs(((x - c_x) * np.cos(t) - ((y - c_y) * np.sin(t))) / (d * s_a)) ** n
               + np.abs(((x - c_x) * np.sin(t) + (y - c_y) * np.cos(t)) / (d * s_b)) ** n)
    cand = np.where(np.max(val, axis=1) < 1)[0]
    t, d, s_a, s_b, c_x, c_y = (q[cand] for q in (t, d, s_a, s_b, c_x, c_y))
    k = np.argmin(4 * d ** 2 * s_a * s_b * gamma(1 + 1 / n) ** 2 / gamma(1 + 2 / n))      # raises on an empty candidate set, as the reference does
    return tuple(q[k][0] for q in (t, d, s_a, s_b, c_x, c_y))


def gen_superellipse_mask(t, d, s_a, s_b, c_x, c_y, n, shape) -> np.ndarray:
    """well_mask_generation.py:94-118"""
    x = np.linspace(-1, 1, shape[0])
    y = np.linspace(-1, 1, shape[1])
    X, Y = np.meshgrid(x, y)
    mask = ((np.abs(((X - c_x) * np.cos(t) - (Y - c_y) * np.sin(t)) / (d * s_a))) ** n
            + (np.abs(((X - c_x) * np.sin(t) + (Y - c_y) * np.cos(t)) / (d * s_b))) ** n < 1)
    return np.swapaxes(mask, 0, 1)


def circle_fallback(shape, mask_val=1) -> np.ndarray:
    """get_circ_mask (:172-182): skimage.draw.disk(center, radius, shape) = points with (r - cr)^2 + (c - cc)^2 < radius^2"""
    m = np.zeros(shape, np.uint8)
    cr, cc = shape[0] // 2, shape[1] // 2
    radius = int(shape[0] * 0.5 * (1 - 0.95))
    rr, cc_ = np.indices(shape)
    m[(rr - cr) ** 2 + (cc_ - cc) ** 2 < radius ** 2] = mask_val
    return m


def generate_well_mask(image, mask_val=1, return_superellipse_params=False, seed=0):
    """well_mask_generation.py:142-233"""
    image = np.asarray(image)
    im_thresh = auto_threshold_well(image)
    ratio = min(1, 200 / np.max(im_thresh.shape))
    small_shape = tuple(int(v) for v in np.round(np.asarray(im_thresh.shape) * ratio))
    im_thresh = resize_nearest(im_thresh, small_shape)
    pts = np.argwhere(border_of(im_thresh))
    try:
        hull = ConvexHull(pts)
    except ValueError:
        return circle_fallback(image.shape, mask_val)
    hv = pts[hull.vertices]
    well = convex_hull_mask(im_thresh.shape, hv)
    wb = border_of(well)
    n = 8 if np.sum(wb) / np.sum(well) > .027 else 2
    x = hv[:, 0] / im_thresh.shape[0] * 2 - 1
    y = hv[:, 1] / im_thresh.shape[1] * 2 - 1
    found = False
    try:
        t, d, s_a, s_b, c_x, c_y = get_superellipse_hull(x, y, n, seed)
        d *= 0.9
        well = gen_superellipse_mask(t, d, s_a, s_b, c_x, c_y, n, im_thresh.shape)
        found = True
    except ValueError:
        pass
    well = resize_nearest(well.astype(np.uint8) * mask_val, image.shape)
    if found and return_superellipse_params:
        return well, t, d, s_a, s_b, c_x, c_y, n
    return well


def make_well_mask(img, seed=0):
    """scripts/compute_branches.py:109-141 -> (well_mask, shrunken_well_mask), boolean"""
    res = generate_well_mask(img, return_superellipse_params=True, seed=seed)
    if isinstance(res, tuple):
        well, t, d, s_a, s_b, c_x, c_y, n = res
        well = well > 0
        shrunk = gen_superellipse_mask(t, d * 0.9, s_a, s_b, c_x, c_y, n, img.shape[:2])
    else:
        well = res > 0
        shrunk = ndi.binary_erosion(well, structure=disk(5), border_value=True)
    if np.sum(well) / well.size < 0.4:
        well = np.ones(img.shape, bool)
        shrunk = np.ones(img.shape, bool)
    return well, shrunk
