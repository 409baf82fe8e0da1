"""oracle/cellarea.py -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU restatement of the cell-area tool (reference scripts/compute_cell_area.py:29-87, 164-178 and
fl_tissue_model_tools/preprocessing.py:44-93): max projection, down-sampling, rescale to 0..1, a two-component gaussian
mixture fitted to the pixel intensities, threshold at foreground mean + sd_coef * foreground sd, area fraction.

The mixture fit belongs to scikit-learn (reference setup.py:70 pins scikit-learn==1.5.0): GaussianMixture(n_components=2,
random_state=rs) = KMeans(2, n_init=1) initialisation + EM (tol 1e-3 on the mean log-likelihood, at most 100 iterations,
reg_covar 1e-6).  sklearn's KMeans draws its k-means++ seeds from the caller's RandomState and stops Lloyd's iteration on
a tolerance, so its labels are not a function of the data alone; what IS a function of the data is the partition it
converges towards.  The restatement used here and on the device is deterministic:
  * initial responsibilities = the GLOBALLY optimal 2-means partition of the intensity histogram (in one dimension: the
    threshold that minimises the within-cluster sum of squares, found by scanning the 65 536 possible thresholds);
  * EM exactly as sklearn/mixture/_gaussian_mixture.py writes it (weights nk / n with nk = sum(resp) + 10 eps, means,
    variances + reg_covar, log-likelihood through the precision, log-sum-exp normalisation), in float64 on the histogram.
tests/test_oracle_cellarea.py pins it two ways: the EM against scikit-learn itself started from the same responsibilities
(agreement to float32 rounding: sklearn runs in float32 on float32 pixels), and the whole threshold against the
reference's own preprocessing.exec_threshold with RandomState(0) (tests/golden/cellarea.npz; tolerance stated there).

cv2.resize is called as cv2.resize(img, dsize, cv2.INTER_AREA) (compute_cell_area.py:57): the third positional parameter
of cv2.resize is `dst`, so the interpolation stays at its default, INTER_LINEAR, and dsize = round(shape * ratio) is read
as (width, height).  OpenCV is absent here: the bilinear kernel below follows the published algorithm (pixel-centre
alignment, float weights for 16-bit input, replicate border, round half to even on the store) -- PARITY UNPINNED.
"""
from __future__ import annotations

import math

import numpy as np

REG_COVAR = 1e-6
TOL = 1e-3
MAX_ITER = 100


def resized_shape(shape, dsamp_size):
    """compute_cell_area.py:54-57: dsize = round(shape * dsamp_size / max(shape)) handed to cv2 as (width, height):
    the result has dsize[1] rows and dsize[0] columns"""
    ratio = dsamp_size / max(shape)
    dsize = tuple(int(v) for v in np.round(np.multiply(shape, ratio)).astype(int))
    return dsize[1], dsize[0]


def linear_axis(n_src, n_dst):
    scale = n_src / n_dst
    i0 = np.zeros(n_dst, np.int64); i1 = np.zeros(n_dst, np.int64)
    w0 = np.zeros(n_dst, np.float32); w1 = np.zeros(n_dst, np.float32)
    for d in range(n_dst):
        fx = np.float32((d + 0.5) * scale - 0.5)        # cv2 computes the source coordinate in float
        sx = int(math.floor(fx))
        fx = np.float32(fx - sx)
        if sx < 0:
            sx, fx = 0, np.float32(0)
        if sx >= n_src - 1:
            sx, fx = n_src - 1, np.float32(0)
        i0[d], i1[d] = sx, min(sx + 1, n_src - 1)
        w0[d], w1[d] = np.float32(1) - fx, fx
    return i0, i1, w0, w1


def resize_linear_u16(img: np.ndarray, out_hw) -> np.ndarray:
    """cv2.resize(INTER_LINEAR) on uint16: horizontal pass then vertical pass in float32, saturating round-half-even store"""
    H, W = img.shape
    oh, ow = out_hw
    if H == 2 * oh and W == 2 * ow:
        # cv::resize switches INTER_LINEAR to the INTER_AREA fast path when both scale factors are exactly 2 (resize.cpp):
        # integer mean of the 2 x 2 block, rounded half up
        a = img.astype(np.uint32)
        return ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint16)
    c0, c1, wc0, wc1 = linear_axis(W, ow)
    r0, r1, wr0, wr1 = linear_axis(H, oh)
    a = img.astype(np.float32)
    rows = a[:, c0] * wc0 + a[:, c1] * wc1                       # float32
    out = rows[r0] * wr0[:, None] + rows[r1] * wr1[:, None]
    return np.clip(np.rint(out), 0, 65535).astype(np.uint16)


def linear_axis_fixed(n_src, n_dst):
    """cv2's coefficient tables for 8-bit sources (imgproc/src/resize.cpp, resizeGeneric_ with fixed_pt): the float weights of
    linear_axis scaled by INTER_RESIZE_COEF_SCALE = 2048 and rounded to short with saturate_cast (cvRound: half to even)"""
    i0, i1, w0, w1 = linear_axis(n_src, n_dst)
    a0 = np.rint(w0.astype(np.float32) * np.float32(2048)).astype(np.int64)
    a1 = np.rint(w1.astype(np.float32) * np.float32(2048)).astype(np.int64)
    return i0, i1, a0, a1


def resize_linear_u8(img: np.ndarray, out_hw) -> np.ndarray:
    """cv2.resize(INTER_LINEAR) on uint8 (OpenCV >= 4.9 < 4.10, setup.py:63; not importable here: published source restated --
    parity unpinned, hand-derived vectors in tests/test_oracle_cellarea.py).  8-bit images take the FIXED-POINT path: 11-bit
    coefficients, horizontal pass into int32 (HResizeLinear<uchar, int, short, 2048>: S[c0] a0 + S[c1] a1), vertical pass
    VResizeLinear<uchar, int, short, FixedPtCast<int, uchar, 22>>'s 8-bit specialisation
        dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
    (the form its SIMD lanes evaluate).  An exact halving on both axes is INTER_AREA's integer mean, as for uint16.  IPP's own
    8-bit linear resize is not used by default (cv::ipp::useIPP_NotExact() is false)."""
    H, W = img.shape
    oh, ow = out_hw
    a = img.astype(np.int64)
    if H == 2 * oh and W == 2 * ow:
        return ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    c0, c1, a0, a1 = linear_axis_fixed(W, ow)
    r0, r1, b0, b1 = linear_axis_fixed(H, oh)
    rows = a[:, c0] * a0 + a[:, c1] * a1                          # int32 in cv2: at most 255 * 2048
    out = (((b0[:, None] * (rows[r0] >> 4)) >> 16) + ((b1[:, None] * (rows[r1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def resize_linear(img: np.ndarray, out_hw) -> np.ndarray:
    """cv2.resize(img, dsize) as the config-5 tools call it (bilinear: see resize_linear_u16), by source dtype: uint8 images go
    through cv2's fixed-point arithmetic, uint16 through its float arithmetic; the result keeps the dtype"""
    if img.dtype == np.uint8:
        return resize_linear_u8(img, out_hw)
    return resize_linear_u16(img.astype(np.uint16), out_hw)


def rescale01(img: np.ndarray) -> np.ndarray:
    """rescale_intensity(img, out_range=(0, 1)).astype(float32) for an integer image (compute_cell_area.py:79)"""
    lo, hi = float(img.min()), float(img.max())
    x = img.astype(np.float64)
    if lo != hi:
        return ((x - lo) / (hi - lo)).astype(np.float32)
    return np.clip(x, 0, 1).astype(np.float32)


def two_means_threshold(levels: np.ndarray, counts: np.ndarray) -> int:
    """index k such that {levels[:k+1]}, {levels[k+1:]} is the 2-means optimum of the weighted points (levels, counts);
    ties take the smallest k"""
    c = counts.astype(np.float64)
    n = np.cumsum(c)
    s = np.cumsum(c * levels)
    N, S = n[-1], s[-1]
    with np.errstate(divide="ignore", invalid="ignore"):
        between = s * s / n + (S - s) ** 2 / (N - n)            # maximising this minimises the within-cluster sum of squares
    ok = (n > 0) & (n < N)
    between = np.where(ok, between, -np.inf)
    return int(np.argmax(between))


def em_fit(levels: np.ndarray, counts: np.ndarray, resp0: np.ndarray):
    """sklearn GaussianMixture(2).fit on the weighted points from initial responsibilities resp0 (n_levels, 2):
    returns (weights, means, variances, n_iter, converged)"""
    x = levels.astype(np.float64)
    c = counts.astype(np.float64)
    n = c.sum()
    eps = np.finfo(np.float64).eps

    def m_step(resp):
        nk = (resp * c[:, None]).sum(0) + 10 * eps
        means = (resp * (c * x)[:, None]).sum(0) / nk
        var = (resp * (c[:, None] * (x[:, None] - means) ** 2)).sum(0) / nk + REG_COVAR
        return nk / n, means, var

    w, mu, var = m_step(resp0)
    lower = -np.inf
    converged = False
    it = 0
    for it in range(1, MAX_ITER + 1):
        prev = lower
        prec_chol = 1.0 / np.sqrt(var)
        log_prob = -0.5 * (math.log(2 * math.pi) + ((x[:, None] - mu) * prec_chol) ** 2) + np.log(prec_chol)
        wl = log_prob + np.log(w)
        mx = wl.max(1)
        lse = mx + np.log(np.exp(wl - mx[:, None]).sum(1))
        lower = float((c * lse).sum() / n)
        resp = np.exp(wl - lse[:, None])
        w, mu, var = m_step(resp)
        if abs(lower - prev) < TOL:
            converged = True
            break
    return w, mu, var, it, converged


def gmm_threshold(img01: np.ndarray, sd_coef: float = 0.0, levels: np.ndarray = None, mask: np.ndarray = None):
    """preprocessing.exec_threshold: (foreground threshold, mask of pixels kept).  `levels`: the integer image img01 was
    rescaled from; the 2-means initialisation then runs on the integer levels (exact integer sums, what the device does)
    instead of on their float32 images -- the same partition up to float32 rounding of the values.  `mask` (the well,
    compute_cell_area.py:81-85): the mixture is fitted to the pixels inside it, pixels outside are 0 (apply_mask)."""
    sel = np.ones(img01.shape, bool) if mask is None else np.asarray(mask).astype(bool)
    vals, counts = np.unique(img01[sel], return_counts=True)
    if levels is not None:
        lv, lc = np.unique(levels[sel], return_counts=True)
        assert len(lv) == len(vals) and np.array_equal(lc, counts)          # the rescale is strictly monotone
        k = two_means_threshold(lv.astype(np.float64), counts)
    else:
        k = two_means_threshold(vals.astype(np.float64), counts)
    resp0 = np.zeros((len(vals), 2))
    resp0[: k + 1, 0] = 1
    resp0[k + 1:, 1] = 1
    w, mu, var, it, conv = em_fit(vals, counts, resp0)
    fg = int(np.argmax(mu))
    thresh = min(255.0, mu[fg] + math.sqrt(var[fg]) * sd_coef)
    masked = np.where(sel, img01, 0)
    kept = np.where(masked <= thresh, 0, masked) > 0
    return thresh, kept


def cell_area(img: np.ndarray, dsamp_size=512, sd_coef=0.0):
    """one image (2-D or Z stack) -> (area fraction, thresholded uint8 image 0 / 255)"""
    if img.ndim == 3:
        img = img.max(0)
    if dsamp_size is not None:
        img = resize_linear(img, resized_shape(img.shape, dsamp_size))
    thresh, kept = gmm_threshold(rescale01(img), sd_coef, levels=img)
    return kept.sum() / kept.size, kept.astype(np.uint8) * 255


def cell_area_well(img: np.ndarray, dsamp_size=512, sd_coef=0.0, seed=0):
    """--detect-well (compute_cell_area.py:117-130, 273-286): the well mask of the down-sampled image, the mixture inside it,
    area relative to the well -> (area fraction, thresholded 0 / 255, well mask 0 / 255)"""
    from . import wellmask
    if img.ndim == 3:
        img = img.max(0)
    if dsamp_size is not None:
        img = resize_linear(img, resized_shape(img.shape, dsamp_size))
    well = np.asarray(wellmask.generate_well_mask(img, mask_val=255, seed=seed))
    inside = well > 0
    thresh, kept = gmm_threshold(rescale01(img), sd_coef, levels=img, mask=inside)
    return kept.sum() / inside.sum(), kept.astype(np.uint8) * 255, well.astype(np.uint8)
