"""oracle/morph.py -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU restatement (numpy + scipy.ndimage) of the pixel stages of analyze_img's 2-D branch
(reference scripts/compute_branches.py:309-361) and of transforms.filter_branch_seg_mask
(reference fl_tissue_model_tools/transforms.py:209-288, 306-361).

The arithmetic of most of these stages lives in third-party packages that are NOT in
/root/reference: opencv-python 4.9 (cv2.resize) and scikit-image 0.22.0 (rescale_intensity,
median, label, regionprops.perimeter, skeletonize, medial_axis, resize), which are themselves
thin layers over scipy.ndimage for everything except the thinning loops.  Each function cites
the call site and restates the published algorithm.  Pins:
  * skeletonize / medial_axis / perimeter / label / filter_branch_seg_mask: golden vectors made
    by running the reference's transforms.py + scikit-image 0.18.3 (the copy in /opt/conda)
    -> tests/golden/filter.npz (tools/make_goldens.py filter);
  * medial_axis tie-break follows 0.18.3's RandomState(0) (0.22.0 is unseeded, i.e. the
    reference itself is not run-to-run deterministic there; SURVEY.md hard part 4);
  * cv2 Lanczos4: "parity unpinned" (no copy of OpenCV in this environment).
"""
from __future__ import annotations

import math

import numpy as np
from scipy import ndimage as ndi

# ------------------------------------------------------------------------------------------
# a1: cv2.resize(img, (w, h), interpolation=cv2.INTER_LANCZOS4)   compute_branches.py:309-312
# ------------------------------------------------------------------------------------------
_S45 = 0.70710678118654752440084436210485
_CS = [(1, 0), (-_S45, -_S45), (0, 1), (_S45, -_S45), (-1, 0), (_S45, _S45), (0, -1), (-_S45, _S45)]


def lanczos4_coeffs(x: float) -> np.ndarray:
    """OpenCV interpolateLanczos4 (imgproc/resize.cpp): 8 taps for fractional offset x, float32."""
    x = np.float32(x)
    y0 = -(float(x) + 3) * math.pi * 0.25
    s0, c0 = math.sin(y0), math.cos(y0)
    co = np.zeros(8, np.float32)
    s = np.float32(0)
    for i in range(8):
        y0_ = np.float32(x + np.float32(3 - i))
        if abs(float(y0_)) >= 1e-6:
            y = -float(y0_) * math.pi * 0.25
            co[i] = np.float32((_CS[i][0] * s0 + _CS[i][1] * c0) / (y * y))
        else:
            co[i] = np.float32(1e30)
        s = np.float32(s + co[i])
    inv = np.float32(np.float32(1.0) / s)
    return (co * inv).astype(np.float32)


def _axis_table(n_src: int, n_dst: int):
    scale = 1.0 / (float(n_dst) / float(n_src))
    idx = np.zeros((n_dst, 8), np.int64)
    co = np.zeros((n_dst, 8), np.float32)
    for d in range(n_dst):
        fx = np.float32((d + 0.5) * scale - 0.5)
        sx = int(math.floor(float(fx)))
        fx = np.float32(fx - np.float32(sx))
        co[d] = lanczos4_coeffs(fx)
        idx[d] = np.clip(np.arange(sx - 3, sx + 5), 0, n_src - 1)
    return idx, co


def lanczos4_resize_u16(img: np.ndarray, out_hw, sat: int = 65535) -> np.ndarray:
    """u16 (H, W) -> u16 (h, w): separable 8-tap Lanczos4, replicate border, f32 accumulate
    left-to-right (no FMA), horizontal then vertical, round-half-even + saturate: cv2's path for CV_16U
    (HResizeLanczos4<ushort, float, float>, Cast<float, ushort>).  uint8 sources take lanczos4_resize_u8."""
    H, W = img.shape
    h, w = out_hw
    xi, xc = _axis_table(W, w)
    yi, yc = _axis_table(H, h)
    src = img.astype(np.float32)
    tmp = np.zeros((H, w), np.float32)
    for k in range(8):
        tmp = (tmp + src[:, xi[:, k]] * xc[None, :, k]).astype(np.float32)
    out = np.zeros((h, w), np.float32)
    for k in range(8):
        out = (out + tmp[yi[:, k], :] * yc[:, k, None]).astype(np.float32)
    return np.clip(np.rint(out), 0, sat).astype(np.uint16)


def lanczos4_resize_u8(img: np.ndarray, out_hw) -> np.ndarray:
    """uint8 sources.  cv2.resize does NOT run the float path on CV_8U: resize.cpp instantiates
    HResizeLanczos4<uchar, int, short> / VResizeLanczos4<uchar, int, short, FixedPtCast<int, uchar, 22>>: the float
    interpolateLanczos4 coefficients become 11-bit fixed point (saturate_cast<short>(c * 2048): round half to even; no
    renormalisation of their sum), both passes accumulate in int32, and the result is (v + 2^21) >> 22 saturated to
    0..255.  `img` holds 8-bit values (any integer dtype); returns uint16 for the common pipeline.
    OpenCV is absent here: parity unpinned against cv2, pinned to hand-derived vectors (tests/test_oracle_morph.py)."""
    img = np.asarray(img)
    if img.max(initial=0) > 255:
        raise ValueError("lanczos4_resize_u8: values above 255")
    H, W = img.shape
    h, w = out_hw
    xi, xc = _axis_table(W, w)
    yi, yc = _axis_table(H, h)
    xa = np.rint(xc * np.float32(2048.0)).astype(np.int64)         # saturate_cast<short>: |c * 2048| < 32768 always
    ya = np.rint(yc * np.float32(2048.0)).astype(np.int64)
    src = img.astype(np.int64)
    tmp = np.zeros((H, w), np.int64)
    for k in range(8):
        tmp += src[:, xi[:, k]] * xa[None, :, k]
    out = np.zeros((h, w), np.int64)
    for k in range(8):
        out += tmp[yi[:, k], :] * ya[:, k, None]
    assert np.abs(out).max(initial=0) < 2 ** 31                    # cv2 accumulates in int: no wrap possible for 8-bit data
    return np.clip((out + (1 << 21)) >> 22, 0, 255).astype(np.uint16)


def lanczos4_resize(img: np.ndarray, out_hw, input_bits: int = 16) -> np.ndarray:
    """cv2.resize(INTER_LANCZOS4) by source depth: float path for uint16, fixed-point path for uint8"""
    return lanczos4_resize_u8(img, out_hw) if input_bits == 8 else lanczos4_resize_u16(img, out_hw)


def target_shape(shape, ratio: float):
    """tuple(np.round(np.multiply(img.shape[:2], ds_ratio)).astype(int))   :309-311 -- the tuple the reference hands to
    cv2.resize as `dsize`.  cv2 reads dsize as (width, height), so the resized array has shape
    resized_shape(shape, ratio) = (target_shape[1], target_shape[0]); the two agree for square images only."""
    return tuple(int(v) for v in np.round(np.multiply(shape[:2], ratio)).astype(int))


def resized_shape(shape, ratio: float):
    """(rows, cols) of cv2.resize(img, target_shape(...)) as the reference calls it   :309-312"""
    ts = target_shape(shape, ratio)
    return (ts[1], ts[0])


# ------------------------------------------------------------------------------------------
# a2 / a17: skimage.exposure.rescale_intensity(img, out_range=(lo, hi))        :316, :419
# ------------------------------------------------------------------------------------------
def rescale_intensity(img: np.ndarray, out_range) -> np.ndarray:
    """integer input -> float64 arithmetic; float32 input -> float32 arithmetic (scalars are weak)."""
    imin, imax = float(img.min()), float(img.max())
    omin, omax = float(out_range[0]), float(out_range[1])
    if np.issubdtype(img.dtype, np.floating):
        ft = img.dtype.type
        x = np.clip(img, ft(imin), ft(imax))
        if imin != imax:
            x = (x - ft(imin)) / ft(imax - imin)
            return x * ft(omax - omin) + ft(omin)
        return np.clip(x, ft(omin), ft(omax))
    x = np.clip(img.astype(np.float64), imin, imax)
    if imin != imax:
        x = (x - imin) / (imax - imin)
        return x * (omax - omin) + omin
    return np.clip(x, omin, omax)


# ------------------------------------------------------------------------------------------
# a12: filter_branch_seg_mask                                             transforms.py:306-361
# ------------------------------------------------------------------------------------------
DISK2 = np.array([[0, 0, 1, 0, 0], [0, 1, 1, 1, 0], [1, 1, 1, 1, 1], [0, 1, 1, 1, 0], [0, 0, 1, 0, 0]], bool)
_EIGHT = np.ones((3, 3), bool)
_CROSS = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], bool)
_SQRT2 = math.sqrt(2.0)


def median13(mask: np.ndarray) -> np.ndarray:
    """skimage.filters.median(mask, footprint=disk(2)) == ndi.median_filter(mode='nearest') (:323)."""
    return ndi.median_filter(mask.astype(bool), footprint=DISK2, mode="nearest")


def perimeter4(region: np.ndarray) -> float:
    """skimage.measure.perimeter(image, neighborhood=4) (regionprops .perimeter, :328)."""
    img = region.astype(np.uint8)
    border = img - ndi.binary_erosion(img, _CROSS, border_value=0).astype(np.uint8)
    code = ndi.convolve(border, np.array([[10, 2, 10], [2, 1, 2], [10, 2, 10]]), mode="constant", cval=0)
    hist = np.bincount(code.ravel(), minlength=50)
    n1 = int(hist[[5, 7, 15, 17, 25, 27]].sum())
    n2 = int(hist[[21, 33]].sum())
    n3 = int(hist[[13, 23]].sum())
    return float(n1) + n2 * _SQRT2 + n3 * ((1 + _SQRT2) / 2)


_SKEL_LUT = np.array([
    0, 0, 0, 1, 0, 0, 1, 3, 0, 0, 3, 1, 1, 0, 1, 3, 0, 0, 0, 0, 0, 0, 0, 0, 2, 0, 2, 0, 3, 0, 3, 3,
    0, 0, 0, 0, 0, 0, 0, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 0, 0, 0, 3, 0, 2, 2,
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    2, 0, 0, 0, 0, 0, 0, 0, 2, 0, 0, 0, 2, 0, 0, 0, 3, 0, 0, 0, 0, 0, 0, 0, 3, 0, 0, 0, 3, 0, 2, 0,
    0, 0, 3, 1, 0, 0, 1, 3, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1,
    3, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    2, 3, 1, 3, 0, 0, 1, 3, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    2, 3, 0, 1, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 3, 3, 0, 1, 0, 0, 0, 0, 2, 2, 0, 0, 2, 0, 0, 0], np.uint8)


def zhang_lut() -> np.ndarray:
    """256-entry table of scikit-image's 2-D skeletonize (`_fast_skeletonize`): neighbour weights
    NW=1, N=2, NE=4, E=8, SE=16, S=32, SW=64, W=128; value 1: removable in the first
    sub-iteration only, 2: second only, 3: both, 0: keep.  It is NOT the textbook Zhang-Suen table
    (corner pixels of 8-connected staircases are removable); the values were recovered from the
    compiled function's behaviour (tools/recover_skel_lut.py) and are pinned by
    tests/golden/filter.npz."""
    return _SKEL_LUT


def skeletonize_zhang(mask: np.ndarray) -> np.ndarray:
    """skimage.morphology.skeletonize(mask) for 2-D input (_fast_skeletonize) (:331)."""
    lut = zhang_lut()
    sk = np.pad(mask.astype(np.uint8), 1)
    wts = np.array([[1, 2, 4], [128, 0, 8], [64, 32, 16]])
    while True:
        removed = False
        for first in (True, False):
            code = ndi.correlate(sk, wts, mode="constant", cval=0)
            val = lut[code]
            kill = (sk > 0) & ((val == 3) | ((val == 1) if first else (val == 2)))
            if kill.any():
                removed = True
                sk = np.where(kill, 0, sk).astype(np.uint8)
        if not removed:
            break
    return sk[1:-1, 1:-1].astype(bool)


def skeleton_components(skel: np.ndarray):
    """nx_graph_from_binary_skeleton (transforms.py:209-288) reduced to what
    filter_branch_seg_mask needs: 8-connected components of the skeleton and, per component,
    whether any node has graph degree > 2 (degree = number of 8-neighbours on the skeleton)."""
    deg = ndi.correlate(skel.astype(np.uint8), np.array([[1, 1, 1], [1, 0, 1], [1, 1, 1]]), mode="constant")
    fork = skel & (deg > 2)
    lab, n = ndi.label(skel, _EIGHT)
    has_fork = np.zeros(n + 1, bool)
    has_fork[np.unique(lab[fork])] = True
    return lab, n, has_fork


def filter_branch_seg_mask(mask: np.ndarray, use_median: bool = True, remove_isolated: bool = True) -> np.ndarray:
    """transforms.py:306-361 (returns the filtered bool mask)."""
    mask = mask.astype(bool)
    if use_median:
        mask = median13(mask)
    else:
        mask = mask.copy()
    lab, n = ndi.label(mask, _EIGHT)
    circ = np.zeros(n + 1)
    for i, sl in enumerate(ndi.find_objects(lab), start=1):
        reg = lab[sl] == i
        per = perimeter4(reg)
        circ[i] = 4 * np.pi * int(reg.sum()) / (per ** 2 + 1e-7)
    skel = skeletonize_zhang(mask)
    slab, sn, has_fork = skeleton_components(skel)
    remove = np.zeros(n + 1, bool)
    for c in range(1, sn + 1):
        ys, xs = np.nonzero(slab == c)
        lbl = lab[ys[0], xs[0]]
        if (remove_isolated and not has_fork[c]) or circ[lbl] > 0.8:
            remove[lbl] = True
    mask[remove[lab]] = False
    return mask


# ------------------------------------------------------------------------------------------
# a14: skimage.morphology.medial_axis(mask, return_distance=True)        compute_branches.py:340
# ------------------------------------------------------------------------------------------
def _pattern_of(index):
    return np.array([[index & 1, index & 2, index & 4], [index & 8, index & 16, index & 32],
                     [index & 64, index & 128, index & 256]], bool)


_MA_TABLE = None


def medial_axis_table():
    global _MA_TABLE
    if _MA_TABLE is None:
        center = (np.arange(512) & 16).astype(bool)
        cond2 = np.array([ndi.label(_pattern_of(i), _EIGHT)[1] != ndi.label(_pattern_of(i & ~16), _EIGHT)[1]
                          for i in range(512)])
        cond3 = np.array([np.sum(_pattern_of(i)) < 3 for i in range(512)])
        _MA_TABLE = (center & (cond2 | cond3)).astype(np.uint8)
    return _MA_TABLE


def medial_axis(mask: np.ndarray):
    """(skeleton bool, distance f64).  Ordered one-pass thinning by
    lexsort((tiebreak, corner_score, distance)); tie-break RandomState(0).permutation (0.18.3)."""
    m = mask.astype(bool)
    table = medial_axis_table()
    dist = ndi.distance_transform_edt(m)
    wts = np.array([[1, 2, 4], [8, 16, 32], [64, 128, 256]])
    code = ndi.correlate(m.astype(np.int32), wts, mode="constant", cval=0)
    corner_tab = np.array([9 - bin(i).count("1") for i in range(512)])
    corner = corner_tab[code]
    ii, jj = np.nonzero(m)
    nfg = len(ii)
    tiebreak = np.random.RandomState(0).permutation(np.arange(nfg))
    order = np.lexsort((tiebreak, corner[m], dist[m]))
    res = np.pad(m.astype(np.int64), 1)
    for k in order:
        i, j = ii[k] + 1, jj[k] + 1
        acc = (16 + res[i - 1, j - 1] + 2 * res[i - 1, j] + 4 * res[i - 1, j + 1] + 8 * res[i, j - 1] + 32 * res[i, j + 1]
               + 64 * res[i + 1, j - 1] + 128 * res[i + 1, j] + 256 * res[i + 1, j + 1])
        res[i, j] = table[acc]
    return res[1:-1, 1:-1].astype(bool), dist


# ------------------------------------------------------------------------------------------
# a15: centre-line weighting                                         compute_branches.py:341-344
# ------------------------------------------------------------------------------------------
def centerline_weight(pred: np.ndarray, mask: np.ndarray):
    skel, dist = medial_axis(mask)
    cdt = ndi.distance_transform_edt(np.logical_not(skel))
    with np.errstate(invalid="ignore", divide="ignore"):
        rel = dist / (dist + cdt)
    return pred * rel, skel, dist


# ------------------------------------------------------------------------------------------
# a16: skimage.transform.resize(order=1, preserve_range=True, anti_aliasing=True)      :351-357
# ------------------------------------------------------------------------------------------
def dsamp_shape(shape, width=384):
    """np.multiply(img.shape[-2:], W / img.shape[-1]).round().astype(int)   :218-222"""
    return tuple(int(v) for v in np.multiply(shape[-2:], width / shape[-1]).round().astype(int))


def gaussian_kernel1d(sigma: float):
    """scipy.ndimage._filters._gaussian_kernel1d(sigma, 0, radius=int(4*sigma+0.5)); exp via libm
    (math.exp) so the C++ product can form bit-identical weights."""
    radius = int(4.0 * float(sigma) + 0.5)
    sigma2 = sigma * sigma
    phi = np.array([math.exp(-0.5 / sigma2 * (x * x)) for x in range(-radius, radius + 1)], np.float64)
    tot = 0.0
    for v in phi:                       # numpy sums < 8 elements sequentially
        tot += v
    assert len(phi) < 8
    return phi / tot


def _mirror(i, n):
    """scipy 'mirror' extension: d c b | a b c d | c b a"""
    if n == 1:
        return 0
    p = 2 * (n - 1)
    i %= p
    return i if i < n else p - i


def correlate1d_sym(a: np.ndarray, w: np.ndarray, axis: int) -> np.ndarray:
    """scipy.ndimage.correlate1d for a symmetric odd kernel, mode='mirror':
    out = x[l]*w[c] + sum_{j=-r..-1} (x[l+j] + x[l-j]) * w[c+j]   (outermost taps first)."""
    r = len(w) // 2
    a = np.moveaxis(a, axis, 0)
    n = a.shape[0]
    idx = lambda off: np.array([_mirror(l + off, n) for l in range(n)])
    out = a * w[r]
    for j in range(-r, 0):
        out = out + (a[idx(j)] + a[idx(-j)]) * w[r + j]
    return np.moveaxis(out, 0, axis)


def zoom_linear_grid(a: np.ndarray, out_shape) -> np.ndarray:
    """scipy.ndimage.zoom(a, order=1, mode='mirror', grid_mode=True) (NI_ZoomShift): source
    coordinate (j + 0.5) * (in/out) - 0.5, weights (1 - t, 1 - (1 - t)), accumulation order
    ((c00*wr0)*wc0 + (c01*wr0)*wc1) + (c10*wr1)*wc0 + (c11*wr1)*wc1."""
    def axis_tab(n_in, n_out):
        zoom = n_in / n_out
        i0 = np.zeros(n_out, np.int64); i1 = np.zeros(n_out, np.int64)
        w0 = np.zeros(n_out); w1 = np.zeros(n_out)
        for j in range(n_out):
            cc = (j + 0.5) * zoom - 0.5
            fl = math.floor(cc)
            t = cc - fl
            w0[j] = 1.0 - t
            w1[j] = 1.0 - w0[j]
            i0[j] = _mirror(int(fl), n_in); i1[j] = _mirror(int(fl) + 1, n_in)
        return i0, i1, w0, w1
    r0, r1, wr0, wr1 = axis_tab(a.shape[0], out_shape[0])
    c0, c1, wc0, wc1 = axis_tab(a.shape[1], out_shape[1])
    t = (a[np.ix_(r0, c0)] * wr0[:, None]) * wc0[None, :]
    t = t + (a[np.ix_(r0, c1)] * wr0[:, None]) * wc1[None, :]
    t = t + (a[np.ix_(r1, c0)] * wr1[:, None]) * wc0[None, :]
    t = t + (a[np.ix_(r1, c1)] * wr1[:, None]) * wc1[None, :]
    return t


def resize_aa(img: np.ndarray, out_shape) -> np.ndarray:
    """scikit-image >= 0.19 resize(order=1, anti_aliasing=True, preserve_range=True):
    ndi.gaussian_filter(sigma=(f-1)/2, mode='mirror') then ndi.zoom(order=1, mode='mirror',
    grid_mode=True), clipped to the input range.  f64 in/out.  Restated tap by tap (and checked
    against scipy itself in tests/test_oracle_morph.py) so the product can match it bit for bit."""
    img = img.astype(np.float64)
    factors = np.divide(img.shape, out_shape)
    sigma = np.maximum(0, (factors - 1) / 2)
    filt = img
    for ax in (0, 1):
        if sigma[ax] > 0:
            filt = correlate1d_sym(filt, gaussian_kernel1d(float(sigma[ax])), ax)
    out = zoom_linear_grid(filt, out_shape)
    return np.clip(out, img.min(), img.max())


def resize_aa_scipy(img: np.ndarray, out_shape) -> np.ndarray:
    """the same through scipy.ndimage directly (what scikit-image calls)"""
    img = img.astype(np.float64)
    factors = np.divide(img.shape, out_shape)
    sigma = np.maximum(0, (factors - 1) / 2)
    filt = ndi.gaussian_filter(img, sigma, cval=0, mode="mirror")
    out = ndi.zoom(filt, [1 / f for f in factors], order=1, mode="mirror", cval=0, grid_mode=True)
    return np.clip(out, img.min(), img.max())


def postprocess(pred: np.ndarray, out_shape):
    """compute_branches.py:334-357 without a well mask: (field f32 out_shape, seg mask, skeleton)."""
    seg = pred > 0.5
    seg = filter_branch_seg_mask(seg)
    weighted, skel, _ = centerline_weight(pred, seg.astype(float))
    field = resize_aa(weighted, out_shape).astype(np.float32)
    return field, seg, skel
